# Regenerates every artefact under profiles/ on an MI355X box (run from the repo root; writes to gpurun_out/).
set -x
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_pytest.txt; cat gpurun_out/final_pytest.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python bench.py --no_overlap > gpurun_out/bench_seq.json 2>/dev/null
python bench.py --use_lfu --no_cpu_baseline > gpurun_out/bench_lfu.json 2>/dev/null
python bench.py --force_sharded --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_sharded_w1.json
python bench.py --async_copy --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_staged.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 64 --warmup 16 --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_torchrun1.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ov -o r01 -- python $R/bench.py --no_cpu_baseline > $R/gpurun_out/prof_ov.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seq -o r01 -- python $R/bench.py --no_cpu_baseline --no_overlap --no_graph > $R/gpurun_out/prof_seq.log 2>&1
mkdir -p $R/gpurun_out/pmc
for m in calib bench; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc/${m}_$c -o p -- python $R/profiles/pmc_probe.py $m > $R/gpurun_out/pmc/${m}_$c.log 2>&1
done; done
cd $R
python profiles/rocpd_summary.py gpurun_out/prof_ov/r01_results.db 40 > gpurun_out/stats_ov.txt
python profiles/rocpd_summary.py gpurun_out/prof_seq/r01_results.db 40 > gpurun_out/stats_seq.txt
python profiles/rocpd_timeline.py gpurun_out/prof_seq/r01_results.db -4 > gpurun_out/timeline_seq.txt
rm -rf gpurun_out/prof_ov gpurun_out/prof_seq
find gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
