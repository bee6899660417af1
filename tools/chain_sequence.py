"""Main-stream kernel SEQUENCE of one training step from a rocprofv3 kernel trace (steps delimited by k_adam): start offset, duration, gap to the previous kernel of the stream.
    python tools/chain_sequence.py <results.db> [first] [count]        (first / count: window of the busiest stream's launches)"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
scol = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name,start,end,{scol},grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
seg = rows[ad[1] + 1:ad[2] + 1]
per = collections.defaultdict(list)
for r in seg:
    per[r[3]].append(r)
main = max(per.values(), key=lambda L: sum(r[2] - r[1] for r in L))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(main)
t0 = main[0][1]
def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(k_\w+?)I(DF16_|DF16b)((?:L[ib]\d+E)*)E", n)
    if m:
        args = ",".join(re.findall(r"L[ib](\d+)E", m.group(3)))
        return m.group(1) + "<" + ("f16" if m.group(2) == "DF16_" else "bf16") + "," + args + ">"
    return re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:58]
gaps = [(main[i][1] - main[i - 1][2]) / 1e3 for i in range(1, len(main))]
print(f"{len(main)} launches on the busiest stream; busy {sum(r[2]-r[1] for r in main)/1e6:.2f} ms; gaps: sum {sum(g for g in gaps if g > 0)/1e3:.2f} ms, median {sorted(gaps)[len(gaps)//2]:.2f} us")
prev = None
for i, r in enumerate(main[first:first + count]):
    gap = (r[1] - prev) / 1e3 if prev is not None else 0.0
    print(f"{first + i:5d} +{(r[1] - t0) / 1e6:8.3f} ms  {(r[2] - r[1]) / 1e3:7.1f} us  gap {gap:6.1f}  {short(r[0]):58s} g=({r[4] // max(r[7], 1)},{r[5]},{r[6]})")
    prev = r[2]
