"""host time per call of the pieces of a prefetch_num = 1 step (Avazu B = 2048 shape), GPU kept far from saturated"""
import sys, time, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
import cachedembedding_amd as ce
from cachedembedding_amd.pipeline import GraphedWindow
torch.manual_seed(0)
N, D, F, B = 9_400_000, 32, 13, 2048
n = F * B
w = ce.HostTable.uniform(N, D, seed=1) if hasattr(ce.HostTable, "uniform") else None
emb = ce.CachedEmbeddingBag(N, D, sparse=True, mode="sum", include_last_offset=True, cache_ratio=0.01, strict=False,
                            evict_strategy=ce.EvictionStrategy.LFU)
emb.set_fused_sgd(0.1); emb.set_cache_op(False)
mgr = emb.cache_weight_mgr
mgr.set_transport("zerocopy")
off = torch.arange(n + 1, dtype=torch.int32, device="cuda")
grad = torch.randn(B, F, D, device="cuda") * 1e-3
ids = [(torch.rand(n, device="cuda") ** 4 * N).long() for _ in range(64)]
out = torch.empty(n, dtype=torch.int64, device="cuda")
def t(fn, reps=40, rounds=20):
    # host enqueue time only: bursts of `reps` calls into an empty queue, the clock stops before the sync
    tot = 0.0
    for r in range(rounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps):
            fn(r * reps + i)
        tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    return tot / (reps * rounds) * 1e6
print("prepare_ids (15 launches + python): %.1f us" % t(lambda i: mgr.prepare_ids(ids[i % 64], out=out)))
import ctypes
from cachedembedding_amd._lib import lib, ptr, stream_ptr, check
h = mgr._handle
def raw(i):
    check(lib.ce_cache_prepare_ids(h, ptr(ids[i % 64]), n, ptr(out), stream_ptr()))
print("ce_cache_prepare_ids via ctypes only: %.1f us" % t(raw))
sp = stream_ptr()
p_ids = [x.data_ptr() for x in ids]; p_out = out.data_ptr()
def raw2(i):
    lib.ce_cache_prepare_ids(h, p_ids[i % 64], n, p_out, sp)
print("ce_cache_prepare_ids, pointers precomputed: %.1f us" % t(raw2))
def step(slots, i, keys=None):
    o = emb(slots, off, hook_features=F); o.backward(grad)
gw = GraphedWindow(emb, 1, n, step, overlap=True, warmup_values=[ids[0]], transport="zerocopy", plan_ahead=2)
print("train graph replay: %.1f us" % t(lambda i: gw._graphs[0].replay()))
def evs(i):
    e = torch.cuda.Event(); e.record(); torch.cuda.current_stream().wait_event(e)
print("Event() + record + wait_event: %.1f us" % t(evs))
print("raise_on_failed_calls: %.1f us" % t(lambda i: mgr.raise_on_failed_calls()))
sub = -1
def full(i):
    gw.submit([ids[i % 64]], i % 3)
    gw.run(i % 3)
print("submit + run (plan_ahead 2 order simplified): %.1f us" % t(full))
