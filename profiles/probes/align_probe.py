"""Does the placement of the forward's output / the backward's gradient tensor matter?  (k_bag_fwd_keys is bimodal over
runs of the same binary: 40.5 or 47-48 us.)  One window of bench-shaped keys, then the two kernels back to back with
the big tensor at different offsets inside one large allocation, and in fresh allocations of different history."""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
import cachedembedding_amd as ce
from cachedembedding_amd import _lib, synthetic
from cachedembedding_amd.functional import presort_window
lib, ptr, sp = _lib.lib, _lib.ptr, _lib.stream_ptr
B, F, D, P = 16384, 26, 128, 8
dev = torch.device("cuda", 0)
sizes = synthetic.TABLES["criteo_1tb"]; N = sum(sizes)
gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(32)
emb = ce.CachedEmbeddingBag(N, D, sparse=True, mode="sum", include_last_offset=True, cache_ratio=0.01, ids_freq_mapping=freq,
                            warmup_ratio=0.7, strict=False)
mgr = emb.cache_weight_mgr; C = mgr.cuda_row_num
for _ in range(12):
    mgr.prepare_ids(gen.next_values(P).view(-1))
off = torch.arange(B * F + 1, dtype=torch.int32, device=dev)
vals = gen.next_values(P)
slots = mgr.prepare_ids(vals.view(-1)).view(P, -1).contiguous()
keys = presort_window(slots, C, offsets=off, include_last_offset=True, hook_features=F, identity_bags=True)
cw = mgr.cuda_cached_weight
n = B * F
def t_fwd(out, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(P): lib.ce_bag_forward_src_keys(cw.data_ptr(), C, D, n, keys[i].keys.data_ptr(), out.data_ptr(), sp())
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        for i in range(P): lib.ce_bag_forward_src_keys(cw.data_ptr(), C, D, n, keys[i].keys.data_ptr(), out.data_ptr(), sp())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * P)
def t_bwd(grad, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(P): lib.ce_bag_backward_sgd_presorted_src(cw.data_ptr(), C, D, n, grad.data_ptr(), 0.0, keys[i].keys.data_ptr(), sp())
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        for i in range(P): lib.ce_bag_backward_sgd_presorted_src(cw.data_ptr(), C, D, n, grad.data_ptr(), 0.0, keys[i].keys.data_ptr(), sp())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * P)
nbytes = n * D * 4
big = torch.empty(nbytes + (512 << 20), dtype=torch.uint8, device=dev)
print("cache table at 0x%x (mod 2 MB: 0x%x), big buffer at 0x%x" % (cw.data_ptr(), cw.data_ptr() % (2 << 20), big.data_ptr()))
for o in [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 64 << 20, 100 << 20, 256 << 20, 500 << 20]:
    v = big[o:o + nbytes].view(torch.float32).view(B, F, D)
    v.normal_()
    print("offset %9d: fwd %.1f us   bwd %.1f us" % (o, t_fwd(v), t_bwd(v)))
for k in range(6):
    junk = torch.empty((k * 37 + 1) << 20, dtype=torch.uint8, device=dev)
    t = torch.randn(B, F, D, device=dev)
    print("fresh alloc %d at 0x%x (mod 2MB 0x%x): fwd %.1f us   bwd %.1f us" % (k, t.data_ptr(), t.data_ptr() % (2 << 20), t_fwd(t), t_bwd(t)))
    del junk
