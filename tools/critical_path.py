"""Where does the wall time of ONE training step go?  From a rocprofv3 kernel trace (steps delimited by k_adam): per 2-ms bin the busy fraction of every stream
and its dominant kernel, and the long idle gaps of the main stream (waits for other streams) with the kernels around them.
    python tools/critical_path.py <results.db> [bin_ms] [gap_us]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
scol = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name,start,end,{scol} from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
seg = rows[ad[1] + 1:ad[2] + 1]
t0, t1 = seg[0][1], seg[-1][2]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:34]
per = collections.defaultdict(list)
for n, s, e, q in seg:
    per[q].append((s, e, short(n)))
order = sorted(per, key=lambda q: -sum(e - s for s, e, _ in per[q]))
print(f"step wall {(t1 - t0) / 1e6:.2f} ms; streams (busiest first): " + ", ".join(f"{q}: {sum(e - s for s, e, _ in per[q]) / 1e6:.1f} ms / {len(per[q])} launches" for q in order))
nb = int((t1 - t0) / 1e6 / bin_ms) + 1
print(f"{'t (ms)':>7s} " + " ".join(f"{'stream ' + str(q):>46s}" for q in order))
for b in range(nb):
    lo, hi = t0 + b * bin_ms * 1e6, t0 + (b + 1) * bin_ms * 1e6
    cells = []
    for q in order:
        busy = 0.0; names = collections.Counter()
        for s, e, n in per[q]:
            o = min(e, hi) - max(s, lo)
            if o > 0:
                busy += o; names[n] += o
        top = names.most_common(1)[0][0] if names else "-"
        cells.append(f"{100 * busy / (bin_ms * 1e6):5.0f}% {top:>39s}")
    print(f"{b * bin_ms:7.1f} " + " ".join(cells))
main = per[order[0]]
print(f"\nidle gaps > {gap_us:.0f} us on the busiest stream ({order[0]}):")
tot = 0.0
for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:]):
    g = (s1 - e0) / 1e3
    if g > gap_us:
        tot += g
        others = [f"{q}:{n}" for q in order[1:] for s, e, n in per[q] if s < s1 and e > e0][:3]
        print(f"  at {(e0 - t0) / 1e6:7.2f} ms: {g:7.1f} us   after {n0:34s} before {n1:34s} meanwhile {others}")
gaps = [(s1 - e0) / 1e3 for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:])]
print(f"  sum of those {tot / 1e3:.2f} ms; all gaps {sum(gaps) / 1e3:.2f} ms over {len(gaps)} boundaries (median {sorted(gaps)[len(gaps) // 2]:.1f} us)")
