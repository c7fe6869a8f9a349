"""Print the per-kernel summary of a rocprofv3 rocpd database (kernel-trace --stats)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print(f"{'kernel':78s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
for n, c, t, a, p in rows[:top]:
    print(f"{n[:78]:78s} {c:6d} {t:12.1f} {a:10.2f} {p:6.2f}")
