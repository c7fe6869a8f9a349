import sys, time, ctypes
sys.path.insert(0, '.')
import torch
import cachedembedding_amd as ce
from cachedembedding_amd._lib import lib, check, ptr, stream_ptr
N, D, C, B, F = 20_000_000, 128, 1_779_442, 16384, 26
emb = ce.CachedEmbeddingBag(N, D, sparse=True, mode="sum", include_last_offset=True, cuda_row_num=C, warmup_ratio=1.0, strict=False)
emb.set_fused_sgd(1.0); emb.set_cache_op(False)
mgr = emb.cache_weight_mgr
off = torch.arange(B * F + 1, dtype=torch.int32, device="cuda")
slots = (torch.rand(B * F, device="cuda") ** 4 * C).long().clamp_(0, C - 1)
grad = torch.randn(B, F, D, device="cuda") * 1e-3
side = torch.cuda.Stream()
def timed(label, side_fn):
    torch.cuda.synchronize()
    if side_fn:
        with torch.cuda.stream(side):
            side_fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fw = bw = 0.0
    n = 20
    for _ in range(n):
        e[0].record(); out = emb(slots, off, hook_features=F); e[1].record(); out.backward(grad); e[2].record()
        torch.cuda.synchronize() if False else None
        e[2].synchronize(); fw += e[0].elapsed_time(e[1]); bw += e[1].elapsed_time(e[2])
    torch.cuda.synchronize()
    print(f"{label:28s} fwd {fw/n*1e3:7.1f} us  bwd {bw/n*1e3:7.1f} us")
timed("warmup", None)
timed("alone", None)
timed("alone again", None)
def reads():
    for _ in range(8):
        check(lib.ce_cache_preload(mgr._handle, None, None, C, stream_ptr()))   # zero-copy host->HBM reads of C rows
def writes():
    for _ in range(4):
        check(lib.ce_cache_flush(mgr._handle, stream_ptr()))
        check(lib.ce_cache_preload(mgr._handle, None, None, C, stream_ptr()))
mgr.flush()
timed("with zero-copy host READS", reads)
torch.cuda.synchronize()
def wr_only():
    for _ in range(4):
        check(lib.ce_cache_flush(mgr._handle, stream_ptr()))
        check(lib.ce_cache_preload(mgr._handle, None, None, 8, stream_ptr()))
timed("with zero-copy host WRITES", wr_only)
