run() { echo "== $*"; timeout 600 python bench.py --no_cpu_baseline "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  %.1f M lookups/s  step %.3f ms steps %d warmup %d' % (d['value']/1e6, d['ms_per_step'], d['steps'], d['warmup']))
except Exception as e:
    print('  FAILED', e)" ; grep -E "Error|error" gpurun_out/err.txt | tail -2; }
run --steps 20 --warmup 5
run --steps 3 --warmup 1
run --steps 100 --warmup 10
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
