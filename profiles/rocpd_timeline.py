"""Timeline of one steady-state window from a rocprofv3 rocpd database: start/end (us, relative) and queue of
every kernel between two consecutive cache-op starts (k_begin, or k_touch -- the per-lookup front has no reset kernel).

    python profiles/rocpd_timeline.py results.db [which]

which = index into the list of k_begin dispatches (negative: from the end; default -3), or the word `steady`: the
middle one of the windows that hold exactly P k_bag_fwd_keys launches (P = the most common count) and no torch
kernel -- i.e. a window of the timed region (graph replays), not one of the eager per-kernel passes behind it."""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
which = sys.argv[2] if len(sys.argv) > 2 else "-3"
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
rows = list(cur.execute("select name, start, end, queue_id from kernels order by start")) if "kernels" in tabs else []
if not rows:
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print(cols)
    sys.exit(0)
begins = [i for i, r in enumerate(rows) if "k_begin" in r[0] or "k_touch" in r[0]]
if which == "steady":
    spans = []
    for k in range(len(begins) - 1):
        seg = rows[begins[k]:begins[k + 1]]
        nf = sum("k_bag_fwd_keys" in r[0] for r in seg)
        torchy = sum("at::native" in r[0] for r in seg) > 2      # (one torch.cat of the window's ids is part of a window)
        spans.append((k, nf, torchy))
    common = Counter(nf for _, nf, t in spans if nf and not t).most_common(1)
    good = [k for k, nf, t in spans if common and nf == common[0][0] and not t]
    k = good[len(good) // 2] if good else len(begins) - 3
    a, b = begins[k], begins[k + 1]
    print(f"# window {k} of {len(begins)} cache ops; {len(good)} windows qualify as steady state")
else:
    w = int(which)
    a, b = begins[w], begins[w + 1]
t0 = rows[a][1]
busy = {}
for name, s, e, q in rows[a:b]:
    short = name.split("(")[0].replace("void ", "").replace("ce::", "")[:38]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} {short}")
    busy[short] = busy.get(short, 0.0) + (e - s) / 1e3
span = (rows[b][1] - t0) / 1e3
print(f"# window span {span:.1f} us; sum of kernel durations by name:")
for k_, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"#   {v:9.1f} {k_}")
