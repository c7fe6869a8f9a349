"""gpurun_out/final5/ (profiles/probes/kinds_smi.sh) -> a table: per process the window time of the pinned side-stream
command beside what rocm-smi showed during the last seconds of that process (clocks, power, temperatures)."""
import json
import re
import sys
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final5")
samples = []                                   # (t, {flattened key: float})
cur_t, buf = None, []


def flush():
    if cur_t is None or not buf:
        return
    try:
        j = json.loads("".join(buf))
    except Exception:
        return
    flat = {}
    for card, kv in j.items():
        if not isinstance(kv, dict):
            continue
        for k, v in kv.items():
            m = re.search(r"-?\d+(\.\d+)?", str(v))
            if m:
                flat[k] = float(m.group(0))
    samples.append((cur_t, flat))


for line in (root / "smi.log").read_text().splitlines():
    if line.startswith("T "):
        flush()
        cur_t, buf = float(line[2:]), []
    elif line.strip():
        buf.append(line)
flush()
keys = sorted({k for _, f in samples for k in f})
want = [k for k in keys if re.search(r"sclk|mclk|fclk|socclk|power|temperature", k, re.I)]
print("# The pinned side-stream command, one process after another, with rocm-smi beside it (round 6)\n")
print(f"{len(samples)} rocm-smi samples; per process: the window of 8 steps, lookups/s, three phase times of the cache op, "
      "and the MAXIMUM of every rocm-smi reading during the process's last 4 s (its trial-free timed region lies there).\n")
runs = [l.split() for l in (root / "runs.txt").read_text().splitlines() if l.strip()]
short = [re.sub(r"\s*\(.*?\)", "", k)[:28] for k in want]
print("| run | ms / window | G lookups/s | find_evict_ids / evict_stage / admit_swap ms | " + " | ".join(short) + " |")
print("|---|---|---|---|" + "---|" * len(want))
for i, s, e in runs:
    f = root / f"run_{i}.json"
    try:
        j = json.loads(f.read_text())
    except Exception:
        print(f"| {i} | (no line) |")
        continue
    ph = j["cache"]["cache_op_ms_by_phase"]
    e = float(e)
    win = [fl for t, fl in samples if e - 4.0 <= t <= e]
    cells = []
    for k in want:
        vals = [fl[k] for fl in win if k in fl]
        cells.append(f"{max(vals):g}" if vals else "-")
    print(f"| {i} | {j['window']['ms']:.3f} | {j['value'] / 1e9:.2f} | {ph['find_evict_ids']:.3f} / {ph['evict_stage']:.3f} / "
          f"{ph['admit_swap']:.3f} | " + " | ".join(cells) + " |")
