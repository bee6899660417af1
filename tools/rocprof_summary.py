"""Text summary (per-kernel calls / total / average / share) of a rocprofv3 `--kernel-trace --stats` sqlite output:
    python tools/rocprof_summary.py <results.db> [title] > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]}")
print(f"# rocprofv3 --kernel-trace --stats; durations in microseconds; total kernel time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
print(f"{'kernel':110s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'share%':>7s}")
for n, c, t, a, p in rows:
    print(f"{n[:110]:110s} {c:7d} {t:12.1f} {a:10.2f} {p:7.2f}")
