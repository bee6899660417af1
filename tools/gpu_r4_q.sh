#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin"
for i in 1 2; do
for lib in playablevideogeneration_amd/csrc/libcaddy_hip.so tools/_ab/libcaddy_hip_prio.so; do echo $lib; CADDY_HIP_LIB=$PWD/$lib timeout 600 $B --no-perceptual 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('erad step ms', d['ms_per_step'])";  CADDY_HIP_LIB=$PWD/$lib timeout 600 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full step ms', d['ms_per_step'])"; done
done
