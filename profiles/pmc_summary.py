"""Per-launch average of one PMC counter for the two bag kernels, from the rocprofv3 --pmc CSVs that
profiles/collect.sh writes (gpurun_out/pmc/<calib|bench>_<COUNTER>/p_counter_collection.csv).
FETCH_SIZE / WRITE_SIZE are reported in KB; the first 2 launches of each kernel are dropped (warm-up)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
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
                for k in ("k_bag_bwd_tile", "k_bag_bwd_stream", "k_bag_fwd", "k_rows_axpy"):
                    if k in r["Kernel_Name"]:
                        vals[k].append(float(r["Counter_Value"]))
        parts = []
        for k, v in sorted(vals.items()):
            v = v[2:] if len(v) > 4 else v
            parts.append(f"{k}: launches={len(v)} avg={sum(v) / len(v):.1f} KB")
        print(f"{mode:5s} {counter:10s} " + "  ".join(parts))
