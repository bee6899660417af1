# environment-variable A/B sweep of the training step on one box (each line: 5 timed steps incl. the VGG19 loss): are the defaults still the best?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region" | sed 's/.*done: //'; echo; }
run X=0
for v in ${SWEEP:-CADDY_BN_SMALL=0 CADDY_FWD_SPLIT=0 CADDY_NARROW=0 CADDY_C4=0 CADDY_WGRAD_TILE=0 CADDY_FIRST_TOUCH=0 CADDY_HX_XCD=0 CADDY_VGG_FUSE_POOL=0 CADDY_WGRAD_HX=0 CADDY_SIDE_STREAM=0}; do run $v; done
run X=1
