#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for lib in tools/_ab/libcaddy_hip_r3.so playablevideogeneration_amd/csrc/libcaddy_hip.so; do echo $lib; CADDY_HIP_LIB=$PWD/$lib BENCH_ONLY=VGG timeout 300 python tools/bench_hx.py 2>&1 | grep VGG | cut -c1-150; done
