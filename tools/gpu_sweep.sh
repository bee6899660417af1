# environment-variable A/B sweep of the training step on one box (each line: 5 timed steps incl. the VGG19 loss)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region" | sed 's/.*done: //'; echo; }
run X=0
run CADDY_WGRAD_BLOCKS=128
run CADDY_WGRAD_BLOCKS=192
run CADDY_WGRAD_BLOCKS=256
run CADDY_WGRAD_BLOCKS=384
run CADDY_HX_BIG=0
run X=1
run CADDY_WGRAD_BLOCKS=256
