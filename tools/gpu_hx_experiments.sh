# timing experiments on conv_hx (results of the experiment builds are numerically WRONG on purpose): which part of the K loop costs what
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
cd playablevideogeneration_amd/csrc
cp libcaddy_hip.so /tmp/libcaddy_hip.so.orig
for e in ${HX_EXPS:-0 1 2 4 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -x hip -I . -I ../../include -DHX_EXPERIMENT=$e -c conv_hx.hip -o /tmp/conv_hx_$e.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcaddy_hip.so $(ls build/*.o | grep -v conv_hx) /tmp/conv_hx_$e.o
  echo "== HX_EXPERIMENT=$e"
  (cd ../..; BENCH_ONLY="${1:-VGG 512->512 @32}" python tools/bench_hx.py 2>&1 | grep TF; BENCH_ONLY="R lstm0" python tools/bench_hx.py 2>&1 | grep TF; BENCH_ONLY="D 128->64" python tools/bench_hx.py 2>&1 | grep TF)
done
cp /tmp/libcaddy_hip.so.orig libcaddy_hip.so
