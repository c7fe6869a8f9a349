timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -x -q 2>&1 | tail -3
for f in "" "--no_overlap"; do timeout 600 python bench.py --force_sharded $f 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('sharded W=1 $f: %.1f M lookups/s  step %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; tail -2 gpurun_out/err.txt | grep -i error; done
