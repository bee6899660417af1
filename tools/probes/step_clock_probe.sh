# engine clock / socket power while the training step loops (is the step power-limited?):  bash tools/probes/step_clock_probe.sh  -> sclk / power samples for the E/R/A/D step, the full step, and an idle baseline
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
sample() { for i in $(seq 1 $1); do sleep 0.7; rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power' | tr -s ' ' | tr '\n' ' '; echo; done; }
echo "idle:"; sample 2
for extra in "--no-perceptual" ""; do
  echo "bench.py --steps 70 $extra:"
  python bench.py --steps 70 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-rollout --no-extra-legs --no-plugin $extra > /tmp/step_clock.json 2> /tmp/step_clock.err &
  pid=$!
  sleep 4.5      # engine creation + warm-up
  sample 6
  wait $pid
  python -c "import json; d = json.loads(open('/tmp/step_clock.json').read().strip().splitlines()[-1]); print('   ms/step', round(d['ms_per_step'], 2))"
done
