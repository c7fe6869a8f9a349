run() { echo "== $*"; timeout 600 python bench.py --no_cpu_baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  %.1f M lookups/s  step %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; }
timeout 600 python -m pytest tests/test_gpu_cache.py -m gpu -x -q 2>&1 | tail -2
run
for sb in 16 32; do echo "swap_blocks=$sb"; CE_SWAP_BLOCKS=$sb run --overlap; done
for mh in 512 2048 8192; do for mb in 256 512 1024; do echo "mark_hot=$mh mark_blocks=$mb"; CE_MARK_HOT=$mh CE_MARK_BLOCKS=$mb run --overlap; done; done
run
run --overlap --use_lfu
