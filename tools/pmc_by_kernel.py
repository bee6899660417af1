"""Issued work per kernel family of one training step from rocprofv3 --pmc passes (--kernel-trace only, a few SQ counters per pass):
    python tools/pmc_by_kernel.py <steps in the profiled command> <db> [<db> ...] > profiles/<round>_issued_work_<workload>.txt
Per family: launches per step and, per step, millions of wave-level instructions by class (VALU incl. MFMA, MFMA alone from SQ_VALU_MFMA_BUSY_CYCLES / 32 for the 32x32x16
instruction -- 16 for the 16x16x32 one, so it is an upper bound there --, SALU, LDS, VMEM) and busy cycles.  Beside the BPTT chain a kernel is paid in the instructions it issues
(profiles/r06_experiments.md): this table is where the E/R/A/D step's issue slots go."""
import collections
import importlib.util
import os
import sqlite3
import sys

spec = importlib.util.spec_from_file_location("pmc_traffic_names", os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.py"))
src = open(spec.origin).read().split("\nfetch, write = collect(")[0]      # the name mapping only
ns = {}
exec(compile(src, spec.origin, "exec"), ns)
short = ns["short"]

steps = int(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counters_collection" in x][0]
    for name, cn, val in c.execute(f"select kernel_name, counter_name, value from {t}"):
        k = short(name)
        agg[k][cn] += val
        cnt[k][cn] += 1
cols = ["SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"]
rows = []
for k, d in agg.items():
    n = max(cnt[k].values())
    rows.append((k, n / steps, [d.get(c_, 0.0) / steps / 1e6 for c_ in cols]))
rows.sort(key=lambda r: -(r[2][0] + r[2][2]))
tot = [sum(r[2][i] for r in rows) for i in range(len(cols))]
print(f"# per training step ({steps} steps profiled); millions of wave-level instructions / cycles")
print(f"{'kernel':44s} {'launches':>8s} {'VALU':>9s} {'MFMA':>9s} {'SALU':>9s} {'LDS':>9s} {'VMEM_RD':>9s} {'VMEM_WR':>9s} {'SQ_BUSY':>10s} {'WAVE_CYC':>11s} {'(V+S-M)/M':>9s}")
for k, n, v in rows:
    mf = v[1] / 32.0
    ratio = (v[0] + v[2] - mf) / mf if mf > 0 else float("nan")
    print(f"{k[:44]:44s} {n:8.0f} {v[0]:9.2f} {mf:9.2f} {v[2]:9.2f} {v[3]:9.2f} {v[4]:9.2f} {v[5]:9.2f} {v[6]:10.1f} {v[7]:11.1f} {ratio:9.1f}")
print(f"{'TOTAL':44s} {sum(r[1] for r in rows):8.0f} {tot[0]:9.2f} {tot[1] / 32:9.2f} {tot[2]:9.2f} {tot[3]:9.2f} {tot[4]:9.2f} {tot[5]:9.2f} {tot[6]:10.1f} {tot[7]:11.1f}")
