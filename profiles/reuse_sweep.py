"""gpurun_out/sweep/ (profiles/reuse_sweep.sh) -> the reuse-sensitivity table (profiles/r04_reuse_sweep.md)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep")
KERNELS = ("k_bag_bwd_stream", "k_bag_fwd_keys", "k_bag_fwd", "k_bag_bwd_tile")


def counters(tag):
    """{kernel: {counter: average KB per launch}} (first 2 launches dropped: warm-up)"""
    out = defaultdict(dict)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = root / "pmc" / f"{tag}_{counter}" / "p_counter_collection.csv"
        if not f.exists():
            continue
        vals = defaultdict(list)
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                for k in KERNELS:
                    if k + "<" in r["Kernel_Name"]:
                        vals[k].append(float(r["Counter_Value"]))
                        break
        for k, v in vals.items():
            v = v[2:] if len(v) > 4 else v
            out[k][counter] = sum(v) / len(v)
    return out


cal = counters("calib")
B, F, D = 16384, 26, 128
known_rd, known_wr = B * F * (512 + 8 + 4) / 1024.0, B * F * 512 / 1024.0
print("# Reuse sensitivity of the bag kernels at the headline shape (B = 16384, F = 26, D = 128, 1 % cache)\n")
print("Regenerate: `bash profiles/reuse_sweep.sh` on an MI355X box, then `python profiles/reuse_sweep.py`.\n")
if not cal:
    print("(`NO_PMC=1`: bench lines only -- kernel times back to back and algorithmic bytes; the counted-traffic columns "
          "need the counter passes, last collected in `r04_reuse_sweep.md`.)\n")
if "k_bag_fwd" in cal:
    c = cal["k_bag_fwd"]
    print(f"Counter calibration (slot-driven forward over 425,984 DISTINCT rows of a 2 GB table: {known_rd:.0f} KB read, "
          f"{known_wr:.0f} KB written per launch by construction): FETCH_SIZE {c.get('FETCH_SIZE', float('nan')):.0f} KB "
          f"= x{known_rd / c['FETCH_SIZE']:.2f} low (gfx950 counts 64 B per 128-B request: MI355X_MICROARCH.md), "
          f"WRITE_SIZE {c.get('WRITE_SIZE', float('nan')):.0f} KB = x{known_wr / c['WRITE_SIZE']:.2f}.  "
          "HBM bytes below = 2 x FETCH_SIZE + WRITE_SIZE.\n")
hdr = ("| id stream | P | distinct rows / batch | lookups/s | kernel | us / launch (back to back) | counted HBM MB | compulsory MB | "
       "SURVEY 8(d) MB | counted / compulsory | counted / 8(d) | frac of 8 TB/s on compulsory | on counted | on 8(d) |")
print(hdr)
print("|" + "---|" * (hdr.count("|") - 1))
for tag, label in (("pl025", "long tail s = 0.25 (default)"), ("mix036", "long tail + 36 % uniform"),
                   ("mix090", "long tail + 90 % uniform"), ("uni", "uniform, Criteo-1TB tables"),
                   ("flat", "uniform, 26 equal tables (no reuse)")):
    f = root / f"{tag}.json"
    if not f.exists() or not f.read_text().strip():
        print(f"| {label} | (bench run failed: see {tag}.err) |")
        continue
    r = json.loads(f.read_text())
    cnt = counters(tag)
    n = B * F
    uniq = r["config"]["distinct_rows_per_batch_frac"] * n
    for o in [r["roofline"]] + r["roofline_other"]:
        k = o["kernel"].split("(")[0]
        if not k.startswith("k_bag"):
            continue
        fwd = k.startswith("k_bag_fwd")
        comp = o["bytes_per_launch"] if fwd else None
        alg = o.get("algorithmic_bytes_per_launch", o["bytes_per_launch"])
        if fwd:
            comp_b, alg_b = comp, alg
        else:           # backward: bytes_per_launch is 8(d)'s figure with the measured unique rows: both the same
            comp_b = alg_b = o["bytes_per_launch"]
        c = cnt.get(k, {})
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 if len(c) == 2 else float("nan")
        us = o["avg_ms"] * 1e3
        fr = lambda b: b / (us * 1e-6) / 8e12
        num = lambda v, fmt: "-" if v != v else format(v, fmt)           # NaN: the counter passes were not run
        print(f"| {label} | {r['config']['prefetch_num']} | {uniq:,.0f} ({100 * uniq / n:.1f} %) | {r['value'] / 1e9:.2f} G | {k} | "
              f"{us:.1f} | {num(hbm / 1e6, '.1f')} | {comp_b / 1e6:.1f} | {alg_b / 1e6:.1f} | {num(hbm / comp_b, '.2f')} | "
              f"{num(hbm / alg_b, '.2f')} | {fr(comp_b):.2f} | {num(fr(hbm), '.2f')} | {fr(alg_b):.2f} |")
