cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for x in 0 1; do
  echo "== CADDY_HX_XCD=$x"
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf gpurun_out/pmc_hx
    CADDY_HX_XCD=$x BENCH_ONLY="VGG 512->512 @32" timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_hx -o hx -- python tools/bench_hx.py > gpurun_out/pmc_hx.log 2>&1
    python - <<'PY'
import sqlite3, collections
c = sqlite3.connect("gpurun_out/pmc_hx/hx_results.db")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name[:75]; agg[k][cn] += val; cnt[(k, cn)] += 1
for k, d in agg.items():
    if "conv_hxIDF16_Li2" not in k: continue
    for cn, v in sorted(d.items()): print(f"   {cn:20s} {v / cnt[(k, cn)]:16.0f} per launch")
PY
  done
  CADDY_HX_XCD=$x BENCH_ONLY="VGG" python tools/bench_hx.py 2>&1 | grep TF | cut -c1-125
  CADDY_HX_XCD=$x timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
done
rm -rf gpurun_out/pmc_hx
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx or folded" 2>&1 | tail -1
