cd $GRAFT_REPO_ROOT
for v in "CADDY_PREZERO=0" "CADDY_PREZERO=1" "CADDY_PREZERO=0" "CADDY_PREZERO=1"; do
  echo "=== $v" >> gpurun_out/ab.txt
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-rollout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('step', d['ms_per_step'], d['loss'])" >> gpurun_out/ab.txt
done
cat gpurun_out/ab.txt
