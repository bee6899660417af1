cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx or folded" 2>&1 | tail -1
for t in 0 1; do
  echo "== CADDY_HX_TALL=$t"
  CADDY_HX_TALL=$t python tools/bench_hx.py 2>&1 | grep -E "64->64|64->32|32->64|128->64" | cut -c1-125
  CADDY_HX_TALL=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region"
  CADDY_HX_TALL=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 --no-perceptual 2>&1 | grep "timed region"
done
