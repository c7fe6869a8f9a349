"""VERDICT r5 #4, the bounded experiment: the key-driven forward on the FASTEST and the SLOWEST of 12 output buffers
of one process, 20 launches each, for a `rocprofv3 --pmc <counter>` pass to count (one process per counter pass:
which buffer is fast is re-established by timing in every process; the last 40 k_bag_fwd_keys dispatches are
20 x fast, then 20 x slow).  Also: the same launch onto `hipMemCreate` memory mapped with the minimum and with the
recommended granularity (profiles/probes/vmm_alloc.cpp is the round-5 form of that).

    python profiles/probes/alloc_pmc.py            -> "fast us / slow us" + the candidates' times on stdout
    python profiles/probes/alloc_pmc.py summary D  -> per-counter averages of the two groups from the CSVs under D"""
import csv
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1].parent
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 2 and sys.argv[1] == "summary":
    root = Path(sys.argv[2])
    print("| counter | fast buffer (avg per launch) | slow buffer | slow / fast |")
    print("|---|---|---|---|")
    for f in sorted(root.glob("*/p_counter_collection.csv")):
        per = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if "k_bag_fwd_keys" in r["Kernel_Name"]:
                    per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, v in per.items():
            v = v[-40:]
            fast, slow = v[:20], v[20:]
            if len(slow) == 20:
                a, b = sum(fast) / 20, sum(slow) / 20
                print(f"| {name} | {a:.4g} | {b:.4g} | {b / a if a else float('nan'):.3f} |")
    sys.exit(0)

import torch  # noqa: E402
from cachedembedding_amd import _lib, synthetic  # noqa: E402
from cachedembedding_amd._lib import check, lib  # noqa: E402
from cachedembedding_amd.functional import presort_window  # noqa: E402

B, F, D, C = 16384, 26, 128, 1_779_442
dev = torch.device("cuda", 0)
n = B * F
gen = synthetic.SyntheticKJT(synthetic.TABLES["criteo_1tb"], B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(16)
rank = torch.empty_like(freq)
rank[torch.argsort(freq, descending=True, stable=True)] = torch.arange(freq.numel(), device=dev)
del freq
off = torch.arange(n + 1, dtype=torch.int32, device=dev)
vals = gen.next_values(4)
slots = (rank[vals] % C).contiguous()
keys = presort_window(slots, C, offsets=off, include_last_offset=True, hook_features=F, identity_bags=True)
del rank
table = torch.randn(C, D, device=dev)
sp = _lib.stream_ptr()


def fwd(o, r=0):
    check(lib.ce_bag_forward_src_keys(table.data_ptr(), C, D, n, keys[r % 4].keys.data_ptr(), o.data_ptr(), sp))


def timed(o, reps=8):
    fwd(o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fwd(o, r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


outs = [torch.empty(n * D, device=dev) for _ in range(12)]
us = [timed(o) for o in outs]
fast, slow = min(range(12), key=lambda i: us[i]), max(range(12), key=lambda i: us[i])
print("candidates us:", " ".join(f"{u:.1f}" for u in us))
print(f"fast = #{fast} {us[fast]:.1f} us @{outs[fast].data_ptr():#x}   slow = #{slow} {us[slow]:.1f} us @{outs[slow].data_ptr():#x}", flush=True)
torch.cuda.synchronize()
for r in range(20):
    fwd(outs[fast], r)
torch.cuda.synchronize()
for r in range(20):
    fwd(outs[slow], r)
torch.cuda.synchronize()
print(f"again: fast {timed(outs[fast]):.1f} us  slow {timed(outs[slow]):.1f} us")
