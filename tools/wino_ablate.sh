#!/bin/bash
# Build timing-ablation variants of K10 (results are garbage, only the duration means something):
#   var/libwino_abl<N>.so for N in "$@"  (bit mask, see DMH_WINO_ABLATE in csrc/wino_conv.hip)
set -e
mkdir -p var
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -DDMH_WINO_ABLATE=$n \
      -shared -o var/libwino_abl$n.so depthmodelhardening_amd/csrc/wino_conv.hip depthmodelhardening_amd/csrc/runtime.hip
done
