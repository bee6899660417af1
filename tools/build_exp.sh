#!/bin/bash
# timing-study builds of the library: conv_hx.hip recompiled with -DHX_EXP=<n> (see conv_hx.hip), everything else from the regular objects -> csrc/build_alt/libexp<n>.so
cd "$(dirname "$0")/../playablevideogeneration_amd/csrc"
mkdir -p build_alt
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -x hip -I . -I ../../include -DHX_EXP=$n -c conv_hx.hip -o build_alt/conv_hx_exp$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/libexp$n.so $(ls build/*.o | grep -v "conv_hx.hip.o") build_alt/conv_hx_exp$n.o -ldl && echo "built libexp$n.so" ) &
done
wait
