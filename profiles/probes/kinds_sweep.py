"""Which PROCESSES get the slow kind of side-stream window (profiles/r06_bench_lines_by_box.md: ~1.4 instead of ~1.05 ms
per window of 8 steps at the headline shape, per process, sticky) -- one bounded experiment: the pinned side-stream
command, a few seconds per process, alternating over
  q8   bench.py as it is (GPU_MAX_HW_QUEUES=8)
  q4 / q16 / q2   another number of hardware queues for HIP to multiplex the streams onto
  hi   the side stream created at the HIGHEST stream priority (its own pool of hardware queues; make_side_stream patched
       in this process only: nothing of the library changes)
Usage (MI355X box, repo root):  python profiles/probes/kinds_sweep.py <rounds> > gpurun_out/kinds_sweep.md
Child mode:  python profiles/probes/kinds_sweep.py --child <priority> <bench args...>"""
import json
import os
import runpy
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
BENCH = ["--arrangement", "overlap", "--steps", "20", "--warmup", "5", "--no_verify", "--no_cpu_baseline"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    prio = sys.argv[2]
    sys.path.insert(0, str(ROOT))
    if prio != "none":
        import torch
        import cachedembedding_amd.pipeline as pl
        lo, hi = -1, 0
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            a, b = ctypes.c_int(), ctypes.c_int()
            hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b))   # (least, greatest): numerically a >= b
            least, greatest = a.value, b.value
        except Exception:
            least, greatest = 0, -1
        want = greatest if prio == "hi" else least
        orig = pl.make_side_stream

        def patched(device, cache_cus=0, total_cus=256):
            if cache_cus > 0:
                return orig(device, cache_cus, total_cus)
            return torch.cuda.Stream(device=device, priority=want)
        pl.make_side_stream = patched
    sys.argv = [str(ROOT / "bench.py")] + sys.argv[3:]
    runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
    sys.exit(0)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CONFIGS = [("q8", {}, "none"), ("q4", {"GPU_MAX_HW_QUEUES": "4"}, "none"), ("q16", {"GPU_MAX_HW_QUEUES": "16"}, "none"),
           ("q2", {"GPU_MAX_HW_QUEUES": "2"}, "none"), ("hi", {}, "hi")]
rows = {c[0]: [] for c in CONFIGS}
for r in range(rounds):
    for name, env, prio in CONFIGS:
        e = dict(os.environ, **env)
        try:
            p = subprocess.run([sys.executable, __file__, "--child", prio] + BENCH, env=e, capture_output=True, text=True,
                               timeout=120, cwd=str(ROOT))
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            ph = j["cache"]["cache_op_ms_by_phase"]
            rows[name].append((j["value"] / 1e9, j["window"]["ms"], ph["find_evict_ids"], ph["evict_stage"], ph["admit_swap"]))
            print(name, r, rows[name][-1], file=sys.stderr, flush=True)          # (progress: the table comes at the end)
        except Exception as ex:                                     # a failed run is a row too
            rows[name].append((float("nan"),) * 5)
            print(f"<!-- {name} round {r}: {type(ex).__name__} {str(ex)[:200]} -->", flush=True)
print("# Side-stream windows by process: hardware-queue count and the side stream's priority (round 6)\n")
print("`python bench.py " + " ".join(BENCH) + "` (the side stream pinned), one process per cell, the configurations taken in "
      "turn; per run: G lookups/s (ms per window of 8 steps; `find_evict_ids` / `evict_stage` / `admit_swap` phase ms).\n")
print("| configuration | " + " | ".join(f"run {i + 1}" for i in range(rounds)) + " | slow kind (> 1.25 ms) |")
print("|---|" + "---|" * (rounds + 1))
for name, _, _ in CONFIGS:
    cells = [f"{v:.2f} ({w:.3f}; {a:.3f} / {b:.3f} / {c:.3f})" for v, w, a, b, c in rows[name]]
    slow = sum(1 for v in rows[name] if v[1] == v[1] and v[1] > 1.25)
    print(f"| {name} | " + " | ".join(cells) + f" | {slow} of {sum(1 for v in rows[name] if v[1] == v[1])} |")
