R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o r01 -- python $R/bench.py --no_cpu_baseline --steps 64 --force_sharded > $R/gpurun_out/prof_x.log 2>&1
cd $R
python profiles/rocpd_timeline.py gpurun_out/prof_x/r01_results.db -4 | awk '{print}' | head -120
rm -rf gpurun_out/prof_x
