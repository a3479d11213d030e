#!/bin/bash
# usage: tools/build_variants.sh NAME="-DFLAG=.. -DFLAG2=.." ...   → build/var/NAME/libptc.so (run locally; the GPU box only benches them)
cd "$(dirname "$0")/../physically-based-renderer_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -I../../include"
n=0
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  mkdir -p ../../build/var/$name
  ( /opt/rocm/bin/hipcc $flags $F -shared -o ../../build/var/$name/libptc.so pt_kernels.hip pt_refit.hip pt_build.hip ptc_api.cpp ptc_scene.cpp > ../../build/var/$name/build.log 2>&1 || echo "build failed: $name" ) &
  n=$((n+1)); if [ $((n % 6)) -eq 0 ]; then wait; fi
done
wait
ls ../../build/var
