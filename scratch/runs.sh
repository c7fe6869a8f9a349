run() { timeout 600 python bench.py --no_cpu_baseline "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  %s -> %.1f M lookups/s  step %.3f ms' % ('$*', d['value']/1e6, d['ms_per_step']))
except Exception as e:
    print('  FAILED', e)" ; grep -E "Error|error|host enqueue" gpurun_out/err.txt | tail -2; }
timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -x -q 2>&1 | tail -2
run --force_sharded
run --force_sharded
