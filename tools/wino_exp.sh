#!/bin/bash
# Builds experiment variants of the Winograd kernel (WINO_EXP bit gates, results are WRONG on purpose -- timing only)
# into tools/exp/libaid_exp<N>.so; run tools/conv_probe.py with AID_LIB_PATH pointing at one of them.
set -e
cd "$(dirname "$0")/.."
S=audio-inpainting-diffusion_amd/csrc; mkdir -p tools/exp
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$S -DWINO_EXP=$v -c $S/aid_conv_wino.hip -o tools/exp/wino_$v.o &
done
wait
for v in "$@"; do
  objs=$(ls $S/build/*.o | grep -v aid_conv_wino)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libaid_exp$v.so $objs tools/exp/wino_$v.o
done
ls -la tools/exp/*.so
