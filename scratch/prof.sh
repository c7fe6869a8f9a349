R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o r01 -- python $R/bench.py --no_cpu_baseline --steps 64 --force_sharded > $R/gpurun_out/prof_x.log 2>&1
cd $R
python profiles/rocpd_timeline.py gpurun_out/prof_x/r01_results.db -4 > gpurun_out/tl.txt
rm -rf gpurun_out/prof_x
python - <<'PY'
rows=[l.split(None,4) for l in open('gpurun_out/tl.txt')]
print('window span us', rows[-1][1])
import collections
busy=collections.defaultdict(float)
for s,e,d,q,n in rows: busy[q]+=float(d)
print('busy per queue', dict(busy))
# main queue = the one with bag kernels
mq=[r for r in rows if 'k_bag' in r[4]][0][3]
prev=None
gaps=[]
for s,e,d,q,n in rows:
    if q!=mq: continue
    if prev is not None and float(s)-prev>30: gaps.append((round(prev,1), round(float(s)-prev,1), n.strip()[:30]))
    prev=float(e)
print('main-queue gaps >30us:', gaps)
agg=collections.defaultdict(float)
for s,e,d,q,n in rows: agg[(q,n.strip()[:28])]+=float(d)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:18]: print(k, round(v,1))
PY
