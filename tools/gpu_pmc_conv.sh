#!/bin/bash
# stall attribution of one conv kernel on one layer shape of tools/bench_step_convs.py (separate --pmc passes, --kernel-trace only):
#   bash tools/gpu_pmc_conv.sh "<BENCH_ONLY substring>" "<kernel name substring>" <out tag>
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
SHAPE="$1"; KSUB="$2"; TAG="$3"
OUT=gpurun_out/pmc_conv_$TAG.txt
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_IFETCH SQ_IFETCH"; do
rm -rf gpurun_out/pmc_c
BENCH_REPS=5 BENCH_ONLY="$SHAPE" timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_c -o c -- python tools/bench_step_convs.py > gpurun_out/pmc_c.log 2>&1 || { echo "set [$set] failed" >> $OUT; grep -i "error\|invalid" gpurun_out/pmc_c.log | head -3 >> $OUT; continue; }
KSUB="$KSUB" python - <<'PY' >> $OUT
import sqlite3, collections, os
c = sqlite3.connect("gpurun_out/pmc_c/c_results.db")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counters_collection" in x][0]
for name, cn, val in c.execute(f"select kernel_name, counter_name, value from {t}"):
    agg[name][cn] += val; cnt[(name, cn)] += 1
for k, d in agg.items():
    if os.environ["KSUB"] not in k: continue
    for cn, v in sorted(d.items()): print(f"{k[:90]:90s} {cn:32s} {v / cnt[(k, cn)]:16.0f} per launch ({cnt[(k, cn)]} launches)")
PY
done
rm -rf gpurun_out/pmc_c
cat $OUT | cut -c60-220
