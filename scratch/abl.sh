for d in 3 4 2 1 0; do echo -n "debug=$d "; CE_BWD_DEBUG=$d timeout 300 python bench.py --no_cpu_baseline --no_overlap --no_graph --steps 64 --warmup 32 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline'] if 'bwd' in d['roofline']['kernel'] else d['roofline_other']; print('  bwd avg_ms %.4f' % r['avg_ms'])"; done
