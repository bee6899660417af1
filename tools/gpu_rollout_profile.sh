# kernel trace of the roll-out (Tennis-main 256x256, S=4, batch 1; eager launches so that every kernel is its own trace record) -> gpurun_out/rollout_kernels.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_roll
CADDY_ROLLOUT_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_roll -o roll -- python tools/bench_rollout.py 36 > gpurun_out/rollout_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_roll/roll_results.db "rollout: 3 x (4 + 36) frames, eager launches" | head -44 > gpurun_out/rollout_kernels.txt
python - <<'PY'
import sqlite3
c = sqlite3.connect("gpurun_out/prof_roll/roll_results.db")
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
rows = c.execute(f"select start, end from {kd} order by start").fetchall()
# idle gaps between consecutive kernels inside the last timed roll-out
rows = rows[-2000:]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
dur = [r[1] - r[0] for r in rows]
import statistics
print(f"last {len(rows)} kernels: median duration {statistics.median(dur) / 1e3:.2f} us, median gap {statistics.median(gaps) / 1e3:.2f} us, sum dur {sum(dur) / 1e6:.2f} ms, sum gaps {sum(g for g in gaps if g < 50000) / 1e6:.2f} ms")
PY
rm -rf gpurun_out/prof_roll
cut -c1-90,105-160 gpurun_out/rollout_kernels.txt | head -44
tail -2 gpurun_out/rollout_prof.log
