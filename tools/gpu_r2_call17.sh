cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
cd playablevideogeneration_amd/csrc
cp libcaddy_hip.so /tmp/libcaddy_hip.so.orig
for e in 0 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -x hip -I . -I ../../include -DHX_EXPERIMENT=$e -c conv_hx.hip -o /tmp/conv_hx_$e.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcaddy_hip.so $(ls build/*.o | grep -v conv_hx) /tmp/conv_hx_$e.o
  echo "== HX_EXPERIMENT=$e"
  (cd ../..; BENCH_ONLY="VGG" python tools/bench_hx.py 2>&1 | grep TF | cut -c1-125; BENCH_ONLY="R lstm" python tools/bench_hx.py 2>&1 | grep TF| cut -c1-125
   timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hx" 2>&1 | tail -1
   timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout --profile-steps 0 2>&1 | grep "timed region")
done
cp /tmp/libcaddy_hip.so.orig libcaddy_hip.so
