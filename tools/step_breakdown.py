"""Per-kernel breakdown of ONE training step from a rocprofv3 kernel trace (steps are delimited by k_adam):
    python tools/step_breakdown.py gpurun_out/prof_serial/bair_results.db [top_n] [--grid]
--grid: one line per (kernel, grid, workgroup) -- i.e. per layer geometry -- instead of per kernel name."""
import collections
import re
import sqlite3
import sys

by_grid = "--grid" in sys.argv
argv = [a for a in sys.argv if a != "--grid"]
c = sqlite3.connect(argv[1])
rows = c.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
seg = rows[ad[1] + 1:ad[2] + 1]
def short(n):
    """mangled k_conv_hx / k_wgrad_hx instances (their _Float16 / __bf16 template argument keeps rocprof from demangling them) -> k_conv_hx<f16, 2, 16, 16, 128, 4, 2, 3, EP, IO>"""
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(k_\w+?)I(DF16_|DF16b)((?:L[ib]\d+E)*)E", n)
    if m:
        args = re.findall(r"L[ib](\d+)E", m.group(3))
        return f"{m.group(1)}<{'f16' if m.group(2) == 'DF16_' else 'bf16'}, {', '.join(args)}>"
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    return n[:70]


agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, gy, gz, wx in seg:
    n = short(n)
    if by_grid:
        n = f"{n[:48]:48s} g=({gx // max(wx, 1)},{gy},{gz}) wg={wx}"
    a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"one step: {tot / 1e3:.2f} ms of kernel time, {len(seg)} launches, wall {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(argv[2]) if len(argv) > 2 else 30]:
    print(f"{n:{80 if by_grid else 62}s} {a[0]:5d} {a[1] / 1e3:8.2f} ms {a[1] / a[0]:8.1f} us  {100 * a[1] / tot:5.1f}%")
