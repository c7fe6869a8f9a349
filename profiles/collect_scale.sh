# Multi-GPU collection for a box with N >= 2 MI355X (the driver's 8-GPU node): the row-wise sharded bench at
# 1/2/4/8 ranks the way the driver launches it, a rocprofv3 kernel trace of the 8-rank run, and the RCCL-backend
# parity test.  Writes to gpurun_out/scale/ (copy the summaries into profiles/ as r02_scale_*).
set -x
R=$PWD
O=$R/gpurun_out/scale
mkdir -p $O
NG=$(python -c "import torch; print(torch.cuda.device_count())")
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -x -q -k rccl 2>&1 | tail -3 > $O/pytest_rccl.txt
python bench.py --gpus 1 --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_n1.json
for n in 2 4 8; do
  [ $n -le $NG ] || continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n \
    bench.py --gpus $n --steps 128 --warmup 32 2> $O/bench_n$n.err | tail -1 > $O/bench_n$n.json
done
cd /tmp; export TMPDIR=/tmp
N=$([ $NG -ge 8 ] && echo 8 || echo $NG)
rocprofv3 --kernel-trace --stats -d $O/prof_n$N -o r02 -- python -m torch.distributed.run --nnodes=1 \
  --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29520 $R/bench.py --gpus $N --steps 64 --warmup 16 \
  > $O/prof_n$N.log 2>&1
cd $R
for db in $O/prof_n$N/*_results.db; do python profiles/rocpd_summary.py $db 30 > $O/stats_$(basename $db .db).txt; done
python - <<'PY'
import json, glob, os
rows = []
for f in sorted(glob.glob(os.path.join(os.environ.get("O", "gpurun_out/scale"), "bench_n*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); rows.append((j["n_gpus"], j["value"], j["ms_per_step"], j["config"].get("prefetch_num")))
    except Exception as e:
        print(f, e)
base = next((v for n, v, *_ in rows if n == 1), None)
for n, v, ms, p in rows:
    print(f"N={n}: {v / 1e9:.3f} G lookups/s, {ms:.3f} ms/step, prefetch_num {p}" + (f", x{v / base:.2f} of N=1" if base else ""))
PY
