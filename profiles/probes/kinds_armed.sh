# The pinned side-stream command WITH the end-of-run check armed (every window's ids stay resident), cut off right after
# its timed region (the line "timed region done" on stderr; the check itself takes 30 s and is not what is asked here):
# does the slow kind of window (r06_kinds_sweep.md: none in 46 processes without the check on one box) come up with it?
BUDGET=${1:-100}
O=gpurun_out/final6; mkdir -p $O
T0=$(date +%s)
while [ $(( $(date +%s) - T0 )) -lt $BUDGET ]; do
  timeout 17 python bench.py --arrangement overlap --steps 20 --warmup 5 --no_cpu_baseline 2>&1 >/dev/null | grep -E "timed region done|warmup done" | cut -c1-220 >> $O/armed.txt
  echo "--" >> $O/armed.txt
done
cat $O/armed.txt
