cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_host_api_gpu.py -m gpu -x -q -k "perceptual or trainer_mirror" 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout > gpurun_out/bench_r2g.json 2> gpurun_out/bench_r2g.err; tail -c 200 gpurun_out/bench_r2g.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2g.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['traffic'], r['traffic_source'], r['frac'])
print({k:(round(v['tflops'],1), round(v['ms_per_step'],2)) for k,v in r['conv_groups'].items()})
PY
