# Regenerates every artefact under profiles/ (r06_*) on an MI355X box (run from the repo root; writes to gpurun_out/).
# NOTE on rocprofv3: with the profiler attached the HIP runtime executes hipMemcpyAsync as a blit KERNEL
# (__amd_rocclr_copyBuffer) instead of an SDMA transfer (profiles/r02_probe_sdma.txt), so a profiled run of the
# default (worker-transport) pipeline shows copy kernels that an unprofiled run does not have, and runs slower.
# Kernel durations for the roofline therefore come from the sequential zero-copy run (every kernel alone on the GPU).
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_pytest.txt; cat gpurun_out/final_pytest.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/bench_driver_args.json 2>/dev/null
python bench.py --no_cpu_baseline --transport zerocopy > gpurun_out/bench_zerocopy.json 2>/dev/null
python bench.py --no_cpu_baseline --no_overlap > gpurun_out/bench_seq.json 2>/dev/null
python bench.py --no_cpu_baseline --use_lfu > gpurun_out/bench_lfu.json 2>/dev/null
python bench.py --no_cpu_baseline --async_copy 2>/dev/null | tail -1 > gpurun_out/bench_staged.json
python bench.py --no_cpu_baseline --unchanged_trainer 2>/dev/null | tail -1 > gpurun_out/bench_unchanged_trainer.json
python bench.py --force_sharded --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_sharded_w1.json
python bench.py --no_cpu_baseline --arrangement interleaved 2>/dev/null | tail -1 > gpurun_out/bench_interleaved.json
python bench.py --no_cpu_baseline --arrangement overlap 2>/dev/null | tail -1 > gpurun_out/bench_overlap.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 64 --warmup 16 --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/bench_torchrun1.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seq -o r06 -- python $R/bench.py --no_cpu_baseline --no_verify --no_overlap --no_graph --transport zerocopy > $R/gpurun_out/prof_seq.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ov -o r06 -- python $R/bench.py --no_cpu_baseline --no_verify > $R/gpurun_out/prof_ov.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_il -o r06 -- python $R/bench.py --no_cpu_baseline --no_verify --arrangement interleaved > $R/gpurun_out/prof_il.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k1 -o r06 -- python $R/bench.py --no_cpu_baseline --no_verify --workload criteo_kaggle --cache_ratio 0.05 --prefetch_num 1 --arrangement interleaved > $R/gpurun_out/prof_k1.log 2>&1
mkdir -p $R/gpurun_out/pmc
for m in calib bench; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc/${m}_$c -o p -- python $R/profiles/pmc_probe.py $m > $R/gpurun_out/pmc/${m}_$c.log 2>&1
done; done
cd $R
python profiles/rocpd_summary.py gpurun_out/prof_seq/r06_results.db 40 > gpurun_out/stats_seq.txt
python profiles/rocpd_summary.py gpurun_out/prof_ov/r06_results.db 40 > gpurun_out/stats_ov.txt
python profiles/rocpd_summary.py gpurun_out/prof_il/r06_results.db 40 > gpurun_out/stats_il.txt
python profiles/rocpd_summary.py gpurun_out/prof_k1/r06_results.db 40 > gpurun_out/stats_k1.txt
python profiles/rocpd_timeline.py gpurun_out/prof_seq/r06_results.db -4 > gpurun_out/timeline_seq.txt
python profiles/rocpd_timeline.py gpurun_out/prof_ov/r06_results.db steady > gpurun_out/timeline_ov.txt
python profiles/rocpd_timeline.py gpurun_out/prof_il/r06_results.db steady > gpurun_out/timeline_il.txt
python profiles/rocpd_timeline.py gpurun_out/prof_k1/r06_results.db steady > gpurun_out/timeline_k1.txt
python profiles/pmc_summary.py gpurun_out/pmc --json gpurun_out/traffic.json > gpurun_out/pmc_hbm_traffic.txt 2>&1
rm -rf gpurun_out/prof_ov gpurun_out/prof_seq gpurun_out/prof_il gpurun_out/prof_k1
find gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
bash profiles/config_matrix.sh > gpurun_out/config_matrix.md 2>&1
# the default lines once more, now that profiles/traffic.json carries this build's digest (roofline.traffic)
cp gpurun_out/traffic.json profiles/traffic.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/bench_driver_args.json 2>/dev/null
bash profiles/dlrm_main_run.sh > gpurun_out/dlrm_main_run.log 2>&1
python profiles/sharded_terms.py 2 4 8 > gpurun_out/sharded_terms.md 2>gpurun_out/sharded_terms.err
for W in 2 3; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2951$W bench.py --gpus $W --share_gpu --verify_sharded --table_scale 0.25 --no_cpu_baseline 2>gpurun_out/share$W.err | tail -1 > gpurun_out/bench_share$W.json
done
