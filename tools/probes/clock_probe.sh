# engine clock / power while one conv_hx shape loops (is the matrix pipe clock-throttled under this kernel?):  bash tools/probes/clock_probe.sh
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for prec in 16 18 0; do BENCH_LOOP_PREC=$prec python tools/probes/clock_loop.py 2>&1 | grep -v amdgpu.ids; done
