#!/bin/bash
# Timing-ablation variants of K18 (results are garbage, only the duration means something): var/libwrw_abl<N>.so for N in "$@"
set -e
mkdir -p var
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -DDMH_WRW_ABLATE=$n \
      -shared -o var/libwrw_abl$n.so depthmodelhardening_amd/csrc/wino_wrw.hip depthmodelhardening_amd/csrc/runtime.hip
done
