#!/bin/bash
# A/B inside ONE gpurun call (box-to-box spread is larger than most effects):  bash tools/gpu_ab.sh "label1|ENV=.. ENV=.." "label2|..." ...   -> gpurun_out/ab.txt
# every run: python bench.py --quick (the contract's timed region + the erad_only leg) with the given environment
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/ab.txt; : > $OUT
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  env $envs timeout 600 python bench.py --steps ${AB_STEPS:-5} --warmup 2 --no-cpu-baseline --profile-steps 0 --quick ${AB_ARGS} > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$label" <<'P' >> $OUT
import json, sys
try:
    d = json.load(open('gpurun_out/ab.json'))
    print(f"{sys.argv[1]:44s} full {d['ms_per_step']:7.2f} ms   erad {d['erad_only']['ms_per_step']:7.2f} ms   loss {d['loss']:.6f}")
except Exception as e:
    print(f"{sys.argv[1]:44s} FAILED {e}"); print(open('gpurun_out/ab.err').read()[-1500:])
P
  tail -1 $OUT
done
