"""Stream-level view of ONE training step from a rocprofv3 kernel trace (steps delimited by k_adam): per stream busy time, idle gaps of the busiest
(critical) stream, and the kernels that occupy it, grouped by a coarse phase guess.   python tools/timeline.py <results.db> [top_n]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("# columns:", cols)
scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = c.execute(f"select name,start,end,{scol or '0'} from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
seg = rows[ad[1] + 1:ad[2] + 1]
t0, t1 = seg[0][1], seg[-1][2]
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(seg)} launches")
per = collections.defaultdict(list)
for n, s, e, q in seg:
    per[q].append((s, e, n))
for q, L in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in L)
    print(f"stream/queue {q}: {len(L)} launches, busy {busy / 1e6:.2f} ms, first {((L[0][0] - t0) / 1e6):.2f} ms, last {((L[-1][1] - t0) / 1e6):.2f} ms")
# union busy time (any stream)
ev = sorted([(s, 1) for _, s, e, _ in seg] + [(e, -1) for _, s, e, _ in seg])
depth = 0; last = t0; idle = 0; multi = 0
for t, d in ev:
    if depth == 0:
        idle += t - last
    if depth >= 2:
        multi += t - last
    depth += d; last = t
print(f"no kernel running: {idle / 1e6:.2f} ms; >= 2 kernels running: {multi / 1e6:.2f} ms")
main = max(per.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in main:
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:60]
    a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"busiest stream: {tot / 1e3:.2f} ms in kernels")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n:62s} {a[0]:5d} {a[1] / 1e3:8.2f} ms {a[1] / a[0]:8.1f} us  {100 * a[1] / tot:5.1f}%")
