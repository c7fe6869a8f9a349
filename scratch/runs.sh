run() { echo "== $*"; timeout 600 python bench.py --no_cpu_baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  %.1f M lookups/s  step %.3f ms  hit %.3f rows_in %d rows_out %d | %s %.3f ms %.0f GB/s | %s %.3f ms %.0f GB/s' % (d['value']/1e6, d['ms_per_step'], d['cache']['unique_hit_rate'], d['cache']['rows_in'], d['cache']['rows_out'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline']['achieved'], d['roofline_other']['kernel'], d['roofline_other']['avg_ms'], d['roofline_other']['achieved']))"; }
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run --use_lfu
run --overlap --use_lfu
run --overlap
