#!/bin/bash
# experiment build of the library: tools/build_exp.sh NAME -DFLAG=... -> tools/exp/libaid_NAME.so (csrc/aid_wino2d.hip recompiled with the flags, every other object from the in-tree build)
# loaded with AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_NAME.so; tools/exp/ is git-ignored and travels to the GPU box
cd "$(dirname "$0")/.."
name=$1; shift
S=audio_inpainting_diffusion_amd/csrc
mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$S -Wno-unused-result -DAID_EXPERIMENT "$@" -c ${EXP_SRC:-$S/aid_wino2d.hip} -o tools/exp/aid_wino2d_$name.o || exit 1
objs=$(ls $S/build/*.hip.o | grep -v aid_wino2d.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libaid_$name.so $objs tools/exp/aid_wino2d_$name.o
