R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ov -o r01 -- python $R/bench.py --no_cpu_baseline --steps 64 > $R/gpurun_out/prof_ov.log 2>&1
cd $R
python profiles/rocpd_timeline.py gpurun_out/prof_ov/r01_results.db -4 > gpurun_out/timeline_ov.txt 2>&1
rm -rf gpurun_out/prof_ov
head -100 gpurun_out/timeline_ov.txt
