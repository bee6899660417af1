#!/bin/bash
# timing-study builds of the library: conv_hx.hip recompiled with extra -D flags, everything else from the regular objects -> csrc/build_alt/lib<name>.so
#   bash tools/build_exp.sh name1:-DHX_EXP=1 name2:-DHX_BG_LOADA_TAP=0 ...      (a bare number n means exp<n>:-DHX_EXP=<n>)
cd "$(dirname "$0")/../playablevideogeneration_amd/csrc"
mkdir -p build_alt
for spec in "$@"; do
  case "$spec" in *:*) name=${spec%%:*}; flags=${spec#*:};; *) name=exp$spec; flags=-DHX_EXP=$spec;; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -x hip -I . -I ../../include $flags -c conv_hx.hip -o build_alt/conv_hx_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/lib$name.so $(ls build/*.o | grep -v "conv_hx.hip.o") build_alt/conv_hx_$name.o -ldl && echo "built lib$name.so" ) &
done
wait
