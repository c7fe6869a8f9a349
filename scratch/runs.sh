run() { timeout 600 python bench.py --no_cpu_baseline "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  %s -> %.1f M lookups/s  step %.3f ms' % ('$*', d['value']/1e6, d['ms_per_step']))
except Exception as e:
    print('  FAILED', e)" ; grep -E "Error|error" gpurun_out/err.txt | tail -2; }
for wb in 0 1 2 4 8; do echo "wb_blocks=$wb"; CE_WB_BLOCKS=$wb run --steps 256; CE_WB_BLOCKS=$wb run --steps 256 --no_overlap; done
