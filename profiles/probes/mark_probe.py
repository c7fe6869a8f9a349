"""k_mark variants at the Criteo-1TB index size (N = 178 M rows, D = 4 so the host table is 2.8 GB), bench-shaped windows
(P = 8 batches of 16384 x 26 ids): the cache op's phases by env setting; one process per setting (the library reads its
switches once).  python profiles/probes/mark_probe.py -> table on stdout"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
import cachedembedding_amd as ce
from cachedembedding_amd import synthetic
dev = torch.device("cuda", 0)
sizes = synthetic.TABLES["criteo_1tb"]; N = sum(sizes); B, P = 16384, 8
gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(32)
emb = ce.CachedEmbeddingBag(N, 4, sparse=True, mode="sum", include_last_offset=True, cache_ratio=0.01, ids_freq_mapping=freq,
                            warmup_ratio=0.7, pin_weight=True, strict=False)
mgr = emb.cache_weight_mgr
wins = [gen.next_values(P).view(-1) for _ in range(24)]
for w in wins[:12]: mgr.prepare_ids(w)
torch.cuda.synchronize()
mgr.set_profiling(True); mgr.phase_times(reset=True)
for w in wins[12:]: mgr.prepare_ids(w)
torch.cuda.synchronize()
ph = mgr.phase_times()
calls = ph.pop("calls")
print(json.dumps({k: round(v / calls * 1e3, 1) for k, v in ph.items()}))
''' % str(ROOT)
variants = [("k_mark (per id)", {"CE_MARK_DEDUPE": "0"}),
            ("dedupe 8192/1024 128KB", {"CE_MARK_DEDUPE": "1"}), ("dedupe 4096/512 64KB", {"CE_MARK_DEDUPE": "2"}),
            ("dedupe 2048/256 32KB", {"CE_MARK_DEDUPE": "3"})]
if os.environ.get("CE_BUILD_ABLATIONS"):
    for name, dd in (("128KB", "1"), ("32KB", "3")):
        for dbg, what in ((8, "no table"), (1, "no idx_map"), (2, "no inverted"), (4, "no bitmap look"), (7, "LDS phase only")):
            variants.append((f"dedupe {name} dbg={dbg} ({what})", {"CE_MARK_DEDUPE": dd, "CE_MARK_DEBUG": str(dbg)}))
for name, env in variants:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    print(f"{name:44s} {line}", flush=True)
