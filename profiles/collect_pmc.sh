# The part of profiles/collect.sh that depends on the exact build: GPU test log, the default bench lines, the
# sequential rocprofv3 stats / timeline and the PMC traffic passes (run from the repo root; writes to gpurun_out/).
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_pytest.txt; cat gpurun_out/final_pytest.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seq -o r04 -- python $R/bench.py --no_cpu_baseline --no_verify --no_overlap --no_graph --transport zerocopy > $R/gpurun_out/prof_seq.log 2>&1
mkdir -p $R/gpurun_out/pmc
for m in calib bench; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc/${m}_$c -o p -- python $R/profiles/pmc_probe.py $m > $R/gpurun_out/pmc/${m}_$c.log 2>&1
done; done
cd $R
python profiles/rocpd_summary.py gpurun_out/prof_seq/r04_results.db 40 > gpurun_out/stats_seq.txt
python profiles/rocpd_timeline.py gpurun_out/prof_seq/r04_results.db -4 > gpurun_out/timeline_seq.txt
python profiles/pmc_summary.py gpurun_out/pmc --json gpurun_out/traffic.json > gpurun_out/pmc_hbm_traffic.txt 2>&1
rm -rf gpurun_out/prof_seq
find gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
# the bench lines read profiles/traffic.json: put the fresh one in place first
cp gpurun_out/traffic.json profiles/traffic.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/bench_driver_args.json 2>/dev/null
python bench.py --force_sharded --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_sharded_w1.json
