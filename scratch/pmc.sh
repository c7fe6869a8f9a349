R=$PWD; cd /tmp; export TMPDIR=/tmp
for m in calib bench; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc/${m}_$c -o p -- python $R/profiles/pmc_probe.py $m > $R/gpurun_out/pmc/${m}_$c.log 2>&1
  echo "$m $c rc=$?"
done; done
find $R/gpurun_out/pmc -name "*.csv" | head -20
