# The pinned side-stream command again and again (one process each, ~9 s) with rocm-smi sampled beside it: do the slower
# and faster windows of profiles/r06_kinds_sweep.md (they drift TOGETHER over tens of seconds, whatever the configuration)
# follow the GPU's clocks, power or temperatures?  Writes gpurun_out/final5/{smi.log, runs.txt, run_*.json}; bounded by
# wall time (first argument, seconds).  profiles/probes/kinds_smi.py turns it into a table.
BUDGET=${1:-230}
O=gpurun_out/final5; mkdir -p $O
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showtemp --showpower --json 2>/dev/null; echo; sleep 0.25; done ) > $O/smi.log &
SMI=$!
T0=$(date +%s)
i=0
while [ $(( $(date +%s) - T0 )) -lt $BUDGET ]; do
  i=$((i+1))
  s=$(date +%s.%N)
  python bench.py --arrangement overlap --steps 20 --warmup 5 --no_verify --no_cpu_baseline 2>$O/err.txt | tail -1 > $O/run_$i.json
  e=$(date +%s.%N)
  echo "$i $s $e" >> $O/runs.txt
done
kill $SMI
wc -l $O/runs.txt; ls -la $O | head -5
