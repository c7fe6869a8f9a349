"""k_touch / k_miss_rank alone on the GPU at the Kaggle P = 1 shape (and a Criteo window): kernel averages from rocprofv3
(one child process per variant, zero-copy transport, 60 calls after 60 warm-up calls) -> profiles/r06_front_probe.txt.

The "default" line needs nothing but the product library.  The ablation lines (CE_TOUCH_DEBUG bits: 1 no idx_map gather,
2 no inverted[] gather, 4 no stamp store, 8 no rows_out store; CE_TOUCH_U / CE_TOUCH_T: ids per thread / threads per
workgroup) read switches the product library does not have: they were compiled into a scratch copy of ce_cache.hip /
ce_cache_fused.h (a `dbg` argument of k_touch and three getenv() calls next to its launch) loaded through
CE_LIBRARY=<that build>; without such a build every line measures the default."""
import json, os, subprocess, sys, sqlite3, glob
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
import cachedembedding_amd as ce
from cachedembedding_amd import synthetic
dev = torch.device("cuda", 0)
wl, P, ratio = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
sizes = synthetic.TABLES[wl]; N = sum(sizes); B = 16384
gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = gen.id_freq_map(32)
emb = ce.CachedEmbeddingBag(N, 4, sparse=True, mode="sum", include_last_offset=True, cache_ratio=ratio, ids_freq_mapping=freq,
                            warmup_ratio=0.7, pin_weight=True, strict=False)
mgr = emb.cache_weight_mgr
wins = [gen.next_values(P).view(-1) for _ in range(120)]
for w in wins[:60]: mgr.prepare_ids(w)
torch.cuda.synchronize()
mgr.set_profiling(True); mgr.phase_times(reset=True)
for w in wins[60:]: mgr.prepare_ids(w)
torch.cuda.synchronize()
ph = mgr.phase_times()
calls = ph.pop("calls")
print(json.dumps({k: round(v / calls * 1e3, 1) for k, v in ph.items()}))
''' % str(ROOT)
open("/tmp/front_child.py", "w").write(CHILD)
def run(name, env, args):
    e = dict(os.environ); e.update(env)
    d = "/tmp/fp_prof"
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "fp", "--", sys.executable, "/tmp/front_child.py", *args],
                       env=e, capture_output=True, text=True, cwd="/tmp")
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    dbs = glob.glob(d + "/**/*.db", recursive=True)
    ks = {}
    if dbs:
        db = sqlite3.connect(dbs[0])
        rows = list(db.execute("select name, start, end from kernels order by start"))
        # last 60 calls only
        for nm in ("k_touch", "k_miss_rank", "k_keys", "k_hist", "k_rank_victims", "k_stage_maps", "k_mark", "k_count", "k_emit", "k_bag_presort", "k_slots", "k_begin"):
            v = [(e_ - s_) / 1e3 for n_, s_, e_ in rows if nm in n_]
            v = v[len(v) // 2:]
            if v: ks[nm] = round(sum(v) / len(v), 1)
    print(f"{name:34s} {ks}  phases {line[-1] if line else r.stderr[-200:]}", flush=True)
lib = os.environ.get("CE_LIBRARY", str(ROOT / "cachedembedding_amd" / "libce_hip.so"))
for wl, P, ratio in (("criteo_kaggle", "1", "0.05"), ("criteo_1tb", "8", "0.01")):
    print(f"## {wl} P={P} ratio={ratio}")
    base = {"CE_LIBRARY": lib, "TMPDIR": "/tmp"}
    for name, env in [("default", {}), ("no idx_map (1)", {"CE_TOUCH_DEBUG": "1"}), ("no inverted (2)", {"CE_TOUCH_DEBUG": "2"}),
                      ("no epoch store (4)", {"CE_TOUCH_DEBUG": "4"}), ("no rows_out (8)", {"CE_TOUCH_DEBUG": "8"}),
                      ("none of them (15)", {"CE_TOUCH_DEBUG": "15"}), ("U=1", {"CE_TOUCH_U": "1"}), ("U=4", {"CE_TOUCH_U": "4"}),
                      ("T=256", {"CE_TOUCH_T": "256"}), ("T=512", {"CE_TOUCH_T": "512"}), ("T=256 U=1", {"CE_TOUCH_T": "256", "CE_TOUCH_U": "1"}),
                      ("old front", {"CE_FRONT_ALL": "0"} if P != "1" else None)]:
        if env is None: continue
        run(name, dict(base, **env), [wl, P, ratio])
