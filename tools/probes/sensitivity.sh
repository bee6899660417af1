#!/bin/bash
# PROBE: where is the step sensitive?  (gpurun)  Whole kernel families switched off in a patched build (tools/probes/build_sensitivity_probe.py) -> gpurun_out/sensitivity.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/sensitivity.txt; : > $OUT
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --no-plugin --profile-steps 0 --quick > gpurun_out/sens.json 2> gpurun_out/sens.err
  python - "$label" <<'P' >> $OUT
import json, sys
try:
    d = json.load(open('gpurun_out/sens.json'))
    print(f"{sys.argv[1]:44s} full {d['ms_per_step']:7.2f} ms   erad {d['erad_only']['ms_per_step']:7.2f} ms")
except Exception as e:
    print(f"{sys.argv[1]:44s} FAILED {e}")
P
  tail -1 $OUT
}
P=tools/_ab/libcaddy_hip_probe.so
N=tools/_ab/libcaddy_hip_probe_noconv.so
run "baseline (in-tree)" A=1
run "probe lib, nothing skipped" CADDY_HIP_LIB=$P
run "skip BN-bwd reduce (lazy chains)" CADDY_HIP_LIB=$P PROBE_SKIP_BN_REDUCE=1
run "skip BN-bwd reduce + apply (lazy chains)" CADDY_HIP_LIB=$P PROBE_SKIP_BN_REDUCE=1 PROBE_SKIP_BN_APPLY=1
run "skip narrow / 1x1 / 7x7 wgrads" CADDY_HIP_LIB=$P PROBE_SKIP_NARROW_WGRAD=1
run "skip k_wgrad_hx" CADDY_HIP_LIB=$P PROBE_SKIP_HX_WGRAD=1
run "skip all wgrads" CADDY_HIP_LIB=$P PROBE_SKIP_HX_WGRAD=1 PROBE_SKIP_NARROW_WGRAD=1
run "no operand conversion in k_conv_hx/k_wgrad_hx" CADDY_HIP_LIB=$N
run "baseline again" A=1
cat $OUT
