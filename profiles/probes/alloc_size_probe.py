"""Follow-up to alloc_lottery.py: is the fast kind of output buffer a matter of HOW MUCH is allocated at once?
(Hypothesis: the page-table fragment of a mapping -- how many bytes one TLB entry covers -- follows the physical
contiguity and the alignment of the allocation; a 208 MiB request is served by 128 + 64 + 16 MiB pieces, a power-of-two
request by one piece.)  Fresh allocations of several sizes (the caching allocator emptied before each), the key-driven
forward timed into the first 208 MiB of each and into a slice further in.
python profiles/probes/alloc_size_probe.py [series | distance | vmm [twice]] -> table on stdout
(vmm: hipcc -shared -fPIC -O2 -o profiles/probes/libvmm_alloc.so profiles/probes/vmm_alloc.cpp first)"""
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
gen = synthetic.SyntheticKJT(synthetic.TABLES["criteo_1tb"], B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(16)
rank = torch.empty_like(freq)
rank[torch.argsort(freq, descending=True, stable=True)] = torch.arange(freq.numel(), device=dev)
del freq
off = torch.arange(n + 1, dtype=torch.int32, device=dev)
vals = gen.next_values(4)
slots = (rank[vals] % C).contiguous()
keys = presort_window(slots, C, offsets=off, include_last_offset=True, hook_features=F, identity_bags=True)
table = torch.randn(C, D, device=dev)
sp = _lib.stream_ptr()


def timed(o, reps=24):
    def fn(r):
        check(lib.ce_bag_forward_src_keys(table.data_ptr(), C, D, n, keys[r % 4].keys.data_ptr(), o.data_ptr(), sp))
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn(r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def tz(x):
    return (x & -x).bit_length() - 1


MiB = 1 << 20


def series(label, count, size=208 * MiB, hold=None):
    out, bufs = [], []
    for _ in range(count):
        b = torch.empty(size // 4, dtype=torch.float32, device=dev)
        bufs.append(b)
        out.append(f"{timed(b[:n * D]):4.1f}@{b.data_ptr() >> 20:x}")
    print(f"{label}: " + " ".join(out), flush=True)
    return bufs


if len(sys.argv) > 1 and sys.argv[1] == "vmm":
    import ctypes
    vmm = ctypes.CDLL(str(Path(__file__).with_name("libvmm_alloc.so")))
    vmm.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    gran = ctypes.c_size_t()
    vmm.vmm_granularity(ctypes.byref(gran))
    print(f"# recommended granularity {gran.value >> 20} MiB; forward us into buffers mapped through hipMemCreate / hipMemMap:")

    class Raw:
        def __init__(self, p):
            self.p = p

        def data_ptr(self):
            return self.p

    def valloc(size, align, piece):
        out = ctypes.c_void_p()
        rc = vmm.vmm_alloc(size, align, piece, ctypes.byref(out))
        assert rc == 0, rc
        return Raw(out.value)

    if len(sys.argv) > 2 and sys.argv[2] == "twice":
        vmm.vmm_alloc_twice.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        print("# the SAME physical pieces (2 MiB each, 208 MiB) at two virtual ranges: forward us into each")
        for shift in (0, 2 * MiB, 6 * MiB, 34 * MiB):
            row = []
            for _ in range(8):
                two = (ctypes.c_void_p * 2)()
                assert vmm.vmm_alloc_twice(208 * MiB, 2 * MiB, shift, two) == 0
                a, b = Raw(two[0]), Raw(two[1])
                row.append(f"{timed(a):4.1f}/{timed(b):4.1f}")
            print(f"second range shifted by {shift >> 20:2d} MiB: " + "  ".join(row), flush=True)
        sys.exit(0)
    size = 256 * MiB
    for align, piece in ((2 * MiB, 2 * MiB), (2 * MiB, 256 * MiB), (256 * MiB, 2 * MiB), (256 * MiB, 32 * MiB),
                         (256 * MiB, 256 * MiB), (1024 * MiB, 256 * MiB), (2 * MiB, 16 * MiB), (32 * MiB, 32 * MiB)):
        row = []
        for _ in range(5):
            o = valloc(size, align, piece)
            row.append(f"{timed(o):4.1f}@{o.data_ptr() >> 20:x}")
        print(f"VA aligned {align >> 20:5d} MiB, pieces of {piece >> 20:4d} MiB: " + " ".join(row), flush=True)
    # the TABLE through the same path (one 1 GiB piece), outputs as torch allocates them
    t2 = valloc(1024 * MiB, 1024 * MiB, 1024 * MiB)
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(t2.data_ptr()), ctypes.c_void_p(table.data_ptr()), ctypes.c_size_t(table.numel() * 4), 3)
    table = t2
    bufs = series("table in ONE 1 GiB-aligned piece, torch outputs", 8)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "distance":
    print("# forward us of fresh 208 MiB outputs allocated BEHIND a ballast of the given size (the table was allocated first)")
    GiB = 1 << 30
    for gb in (0, 1, 2, 4, 8, 16, 32, 48, 64, 80, 96, 112, 128, 160, 192, 224, 0):
        torch.cuda.empty_cache()
        ballast = torch.empty(gb * (GiB // 4), dtype=torch.float32, device=dev) if gb else None
        a = series(f"ballast {gb:3d} GiB", 6)
        del a, ballast
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "series":
    print("# forward us @ address (MiB, hex) of consecutive fresh 208 MiB allocations, all kept alive within a series")
    torch.cuda.empty_cache()
    a = series("A: 24 in a row        ", 24)
    del a
    torch.cuda.empty_cache()
    a = series("B: freed, 24 again    ", 24)
    del a
    torch.cuda.empty_cache()
    ballast = torch.empty(8 << 28, dtype=torch.float32, device=dev)       # 8 GiB
    a = series("C: behind 8 GiB       ", 12)
    del a, ballast
    torch.cuda.empty_cache()
    a = series("D: 211 MiB each       ", 12, size=211 * MiB)
    del a
    torch.cuda.empty_cache()
    a = series("E: 416 MiB each       ", 8, size=416 * MiB)
    sys.exit(0)

keep = []
print("# size of the allocation -> per fresh allocation: log2 alignment of its address, forward us into bytes [0, 208 MiB), "
      "and into a slice starting 256 MiB in (where it fits)")
for size in (208 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB, 2048 * MiB, 208 * MiB):
    row = []
    for _ in range(4):
        torch.cuda.empty_cache()
        r0 = torch.cuda.memory_reserved(dev)
        buf = torch.empty(size // 4, dtype=torch.float32, device=dev)
        keep.append(buf)
        fresh = torch.cuda.memory_reserved(dev) > r0
        t0 = timed(buf[:n * D])
        t1 = timed(buf[64 * MiB:64 * MiB + n * D]) if size >= 512 * MiB else float("nan")
        row.append(f"2^{tz(buf.data_ptr())}{'' if fresh else '(cached)'} {t0:5.1f} {t1:5.1f}")
    print(f"{size // MiB:5d} MiB: " + " | ".join(row))
