"""Two kinds of side-stream window: does running one-stream windows in between flip the side stream into its fast kind?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
import cachedembedding_amd as ce
from cachedembedding_amd import synthetic
from cachedembedding_amd.pipeline import GraphedWindow

dev = torch.device("cuda", 0)
sizes = synthetic.TABLES["criteo_1tb"]; N = sum(sizes); F = len(sizes); B, D, P, L = 16384, 128, 8, 1
gen = synthetic.SyntheticKJT(sizes, B, L, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(sample_batches=4 * P)
embed = ce.CachedEmbeddingBag(N, D, sparse=True, mode="sum", include_last_offset=True, cache_ratio=0.01, ids_freq_mapping=freq,
                              warmup_ratio=0.7, pin_weight=True, evict_strategy=ce.EvictionStrategy.DATASET, init_seed=1024, strict=False)
del freq
mgr = embed.cache_weight_mgr
mgr.set_transport("worker")
embed.set_fused_sgd(1.0)
embed.set_cache_op(False)
while mgr.cuda_available_row_num > 0:
    mgr.prepare_ids(gen.next_values(P).view(-1))
offsets = gen.offsets
grad = torch.randn(B, F, D, device=dev) * 1e-3
grad -= grad.mean(dim=0, keepdim=True)
layout = (offsets, embed.include_last_offset, F)
NW = 1000
wins = [gen.next_values(P) for _ in range(NW)]

def train_step(slots_i, i, keys_i=None):
    out = embed(slots_i, offsets, hook_features=F, presorted=keys_i)
    out.backward(grad)

mode0 = sys.argv[1] if len(sys.argv) > 1 else "overlap"
gw = GraphedWindow(embed, P, B * F * L, train_step, overlap=True, warmup_values=[wins[0][i] for i in range(P)], presort=True,
                   transport=None, bag_layout=layout, plan_ahead=1, arrangement=mode0)
state = {"w": 1}
gw.submit([wins[1][j] for j in range(P)], 1 % 2)

def run_windows(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        w = state["w"]
        gw.submit([wins[w + 1][j] for j in range(P)], (w + 1) % 2)
        gw.run(w % 2)
        state["w"] = w + 1
    e1.record()
    return e0, e1, n

plan = sys.argv[2].split(",") if len(sys.argv) > 2 else ["o32", "o32", "o32", "i1", "o32", "o32", "i8", "o32", "o32", "i32", "o32", "o32"]
blocks = []
for item in plan:
    mode = "overlap" if item[0] == "o" else "interleaved"
    if gw.arrangement != mode:
        gw.set_arrangement(mode)
    blocks.append((item, run_windows(int(item[1:]))))
gw.drain()
torch.cuda.synchronize()
print("start", mode0, " ".join("%s:%.3f" % (it, e0.elapsed_time(e1) / n) for it, (e0, e1, n) in blocks), flush=True)
