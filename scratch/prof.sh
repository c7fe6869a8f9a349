R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ov -o r01 -- python $R/bench.py --no_cpu_baseline > $R/gpurun_out/prof_ov.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seq -o r01 -- python $R/bench.py --no_cpu_baseline --no_overlap > $R/gpurun_out/prof_seq.log 2>&1
cd $R
python profiles/rocpd_summary.py gpurun_out/prof_ov/r01_results.db 45 > gpurun_out/stats_ov.txt
python profiles/rocpd_summary.py gpurun_out/prof_seq/r01_results.db 45 > gpurun_out/stats_seq.txt
grep -E "^\{" gpurun_out/prof_ov.log | cut -c1-200; grep -E "^\{" gpurun_out/prof_seq.log | cut -c1-200
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json | cut -c1-3000
