# stall attribution of k_conv_hx on one layer shape (separate --pmc passes, --kernel-trace only):  bash tools/gpu_pmc_hx.sh ["VGG 512->512 @32"]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
SHAPE="${1:-VGG 512->512 @32}"
: > gpurun_out/pmc_hx_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
rm -rf gpurun_out/pmc_hx
BENCH_ONLY="$SHAPE" timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_hx -o hx -- python tools/bench_hx.py > gpurun_out/pmc_hx.log 2>&1 || { echo "set [$set] failed"; grep -i "error\|invalid\|not" gpurun_out/pmc_hx.log | head -3; continue; }
python - <<'PY' | tee -a gpurun_out/pmc_hx_counters.txt
import sqlite3, collections
c = sqlite3.connect("gpurun_out/pmc_hx/hx_results.db")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name[:75]; agg[k][cn] += val; cnt[(k, cn)] += 1
for k, d in agg.items():
    if "conv_hxIDF16_Li2" not in k: continue
    for cn, v in sorted(d.items()): print(f"{k[-40:]}   {cn:32s} {v / cnt[(k, cn)]:16.0f} per launch")
PY
done
rm -rf gpurun_out/pmc_hx
