"""Per-launch average of one PMC counter for the two bag kernels, from the rocprofv3 --pmc CSVs that
profiles/collect.sh writes (gpurun_out/pmc/<calib|bench>_<COUNTER>/p_counter_collection.csv).
FETCH_SIZE / WRITE_SIZE are reported in KB; the first 2 launches of each kernel are dropped (warm-up)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
# --json FILE: also write the per-launch HBM traffic of the bench kernels (2 x FETCH_SIZE + WRITE_SIZE, the
# calibration of the `calib` pass) as profiles/traffic.json, stamped with the source digest of the library that ran
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
avgs = {}
for mode in ("calib", "bench"):
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = root / f"{mode}_{counter}" / "p_counter_collection.csv"
        if not f.exists():
            continue
        vals = defaultdict(list)
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                for k in ("k_bag_bwd_tile", "k_bag_bwd_stream", "k_bag_fwd_keys", "k_bag_fwd", "k_rows_axpy"):
                    if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:      # (k_bag_fwd is a prefix of k_bag_fwd_keys)
                        vals[k].append(float(r["Counter_Value"]))
                        break
        parts = []
        for k, v in sorted(vals.items()):
            v = v[2:] if len(v) > 4 else v
            parts.append(f"{k}: launches={len(v)} avg={sum(v) / len(v):.1f} KB")
            avgs[(mode, counter, k)] = sum(v) / len(v)
        print(f"{mode:5s} {counter:10s} " + "  ".join(parts))

if json_out:
    import json
    repo = Path(__file__).resolve().parents[1]
    stamp = (repo / "cachedembedding_amd" / "csrc" / ".build_stamp").read_text().strip()
    names = {"k_bag_fwd": "k_bag_fwd", "k_bag_fwd_keys": "k_bag_fwd_keys", "k_bag_bwd_stream": "k_bag_bwd_stream(sgd)",
             "k_bag_bwd_tile": "k_bag_bwd_tile(sgd)"}
    entry = {}
    for k, label in names.items():
        f, w = avgs.get(("bench", "FETCH_SIZE", k)), avgs.get(("bench", "WRITE_SIZE", k))
        if f is not None and w is not None:
            entry[label] = (2.0 * f + w) * 1024.0
    cal = {c: avgs.get(("calib", c, "k_bag_fwd")) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    out = {"criteo_1tb:B16384:D128": entry, "_build_stamp": stamp,
           "_calibration_KB": cal,
           "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (profiles/collect.sh); bytes = "
                      "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 64 B per 128-B request: "
                      "MI355X_MICROARCH.md; confirmed by the calib pass on a no-reuse gather of known size)"}
    Path(json_out).write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", json_out)
