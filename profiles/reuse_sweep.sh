# Reuse sensitivity of the two bag kernels at the headline batch shape (B = 16384, F = 26, D = 128, 1 % cache):
# bench.py lines + rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, --kernel-trace only: the
# MI355X_MICROARCH.md recipe) for id streams from the long-tail default to no reuse at all.  prefetch_num is lowered
# where a window of 8 batches no longer fits the cache.  Run from the repo root on an MI355X box; writes
# gpurun_out/sweep/, which profiles/reuse_sweep.py turns into profiles/r0N_reuse_sweep.md.
# NO_PMC=1: the bench lines only (kernel times and algorithmic bytes; the counter passes take ~10 GPU-minutes and
# their counted / compulsory ratios are a property of the kernels, which have not changed since round 4).
set -x
R=$PWD
O=$R/gpurun_out/sweep
mkdir -p $O/pmc
run() {   # tag workload dist skew uniform_frac prefetch_num
  python bench.py --workload $2 --dist $3 --skew $4 --uniform_frac $5 --prefetch_num $6 --no_verify --no_cpu_baseline \
      --steps 64 --warmup 8 > $O/$1.json 2> $O/$1.err || tail -3 $O/$1.err
  [ -n "$NO_PMC" ] || ( cd /tmp; export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$1_$c -o p -- \
          python $R/profiles/pmc_probe.py bench $2 $3 $4 $5 $6 > $O/pmc/$1_$c.log 2>&1
    done )
}
[ -n "$NO_PMC" ] || ( cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/calib_$c -o p -- python $R/profiles/pmc_probe.py calib > $O/pmc/calib_$c.log 2>&1
  done )
run pl025  criteo_1tb power_law 0.25 0    8
run mix036 criteo_1tb power_law 0.25 0.36 4
run mix090 criteo_1tb power_law 0.25 0.9  2
run uni    criteo_1tb uniform   0.25 0    2
run flat   flat_178m  uniform   0.25 0    1
find $O/pmc -name "*kernel_trace.csv" -delete
find $O/pmc -name "*.db" -delete
du -sh $O
