for d in 0 5; do echo "debug=$d"; CE_BWD_DEBUG=$d timeout 300 python bench.py --no_cpu_baseline --steps 64 --warmup 96 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline'] if 'bwd' in d['roofline']['kernel'] else d['roofline_other']; print('  bwd avg_ms', r['avg_ms'], 'step ms', d['ms_per_step'], d['value']/1e6)"; done
