run() { echo "== $*"; timeout 600 python bench.py --no_cpu_baseline "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  %.1f M lookups/s  step %.3f ms | %s %.4f ms | %s %.4f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline_other']['kernel'], d['roofline_other']['avg_ms']))
except Exception as e:
    print('  FAILED', e)" ; grep -E "Error|error" gpurun_out/err.txt | tail -2; }
timeout 600 python -m pytest tests/test_gpu_bag.py tests/test_gpu_cache.py -m gpu -x -q 2>&1 | tail -2
run --no_overlap --no_graph
run
run
run --pooling 8 --batch_size 4096
