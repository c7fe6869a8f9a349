"""Does the key-driven forward's time depend on WHERE its output (and the table) lie?  Across processes the same launch
measures 39-40 or 46-47 us back to back ("bimodal").  This probe times the same keys against several
output buffers, table copies and gradient buffers inside ONE process; if the spread shows up here, a window object could
try a few allocations when it captures its steps and keep the fast one.
python profiles/probes/alloc_lottery.py -> table on stdout"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1].parent
sys.path.insert(0, str(ROOT))
from cachedembedding_amd import _lib, synthetic  # noqa: E402
from cachedembedding_amd._lib import check, lib  # noqa: E402
from cachedembedding_amd.functional import presort_window  # noqa: E402

B, F, D, C = 16384, 26, 128, 1_779_442
dev = torch.device("cuda", 0)
n = B * F
early = [torch.empty(n * D, device=dev) for _ in range(3)]        # allocated before anything else lives on the device
gen = synthetic.SyntheticKJT(synthetic.TABLES["criteo_1tb"], B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(16)
rank = torch.empty_like(freq)
rank[torch.argsort(freq, descending=True, stable=True)] = torch.arange(freq.numel(), device=dev)
del freq
off = torch.arange(n + 1, dtype=torch.int32, device=dev)
vals = gen.next_values(4)
slots = (rank[vals] % C).contiguous()                       # hot rows -> low slots, the tail spread over the cache
keys = presort_window(slots, C, offsets=off, include_last_offset=True, hook_features=F, identity_bags=True)
sp = _lib.stream_ptr()


def timed(fn, reps=24):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn(r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tables = [torch.randn(C, D, device=dev) for _ in range(3)]
arena = torch.empty(6 * n * D + (64 << 20), dtype=torch.float32, device=dev)
assert arena.data_ptr() % (2 << 20) == 0
# slices of one allocation at offsets (in floats) whose BYTE offset is: 0, 4 KB, 1 KB, 0xe400, 0xde400, 0xede400 past a 2 MB line
offs_elems = [0, n * D + 1024, 2 * n * D + 256, 3 * n * D + 0xe400 // 4, 4 * n * D + 0xde400 // 4, 5 * n * D + 0xede400 // 4]
outs = [arena[o:o + n * D] for o in offs_elems] + [torch.empty(n * D, device=dev) for _ in range(3)] + early
from cachedembedding_amd.functional import probe_rows  # noqa: E402
print("# ce_probe_rows (write us, read us) per output buffer [6 arena slices, 3 late allocations, 3 EARLY allocations]:")
print("  " + "  ".join("%.1f/%.1f" % probe_rows(o.view(n, D), F) for o in outs))
print("# forward from keys: us per launch (24 back to back), by table copy x output buffer")
for ti, t in enumerate(tables):
    row = []
    for oi, o in enumerate(outs):
        us = timed(lambda r: check(lib.ce_bag_forward_src_keys(t.data_ptr(), C, D, n, keys[r % 4].keys.data_ptr(), o.data_ptr(), sp)))
        row.append(us)
    print(f"table {ti} @{t.data_ptr() % (1 << 30):#011x}: " + "  ".join(f"{u:5.1f}" for u in row))
print("# output buffers @ (mod 1 GiB): " + " ".join(f"{o.data_ptr() % (1 << 30):#x}" for o in outs))
grads = [torch.randn(n * D, device=dev) * 1e-3 for _ in range(4)] + [arena[o:o + n * D] for o in offs_elems[:3]] + early
print("# streaming backward + SGD: us per launch, by table copy x gradient buffer")
for ti, t in enumerate(tables[:2]):
    row = []
    for g in grads:
        us = timed(lambda r: check(lib.ce_bag_backward_sgd_presorted_src(t.data_ptr(), C, D, n, g.data_ptr(), 1e-6,
                                                                         keys[r % 4].keys.data_ptr(), sp)))
        row.append(us)
    print(f"table {ti}: " + "  ".join(f"{u:5.1f}" for u in row))
