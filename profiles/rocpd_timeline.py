"""Timeline of one steady-state window from a rocprofv3 rocpd database: start/end (us, relative) and queue of
every kernel between two consecutive k_begin dispatches."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
rows = list(cur.execute("select name, start, end, queue_id from kernels order by start")) if "kernels" in tabs else []
if not rows:
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print(cols)
    sys.exit(0)
begins = [i for i, r in enumerate(rows) if "k_begin" in r[0]]
a, b = begins[which], begins[which + 1]
t0 = rows[a][1]
for name, s, e, q in rows[a:b]:
    short = name.split("(")[0].replace("void ", "").replace("ce::", "")[:38]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} {short}")
