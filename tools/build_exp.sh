#!/bin/bash
# experiment build of the library: tools/build_exp.sh NAME -DFLAG=... -> tools/exp/libaid_NAME.so (one source of csrc/ recompiled with the flags -- EXP_FILE, default
# aid_wino2d.hip; EXP_SRC = a patched copy to compile instead, e.g. after `patch -o tools/exp/x.hip csrc/aid_wino2d.hip tools/r06_c96_three_wave_unfolded.patch` --
# every other object from the in-tree build).  Loaded with AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_NAME.so; tools/exp/ is git-ignored and travels to the GPU box
cd "$(dirname "$0")/.."
name=$1; shift
S=audio_inpainting_diffusion_amd/csrc
F=${EXP_FILE:-aid_wino2d.hip}
mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$S -Wno-unused-result -DAID_EXPERIMENT "$@" -c ${EXP_SRC:-$S/$F} -o tools/exp/${F%.hip}_$name.o || exit 1
objs=$(ls $S/build/*.hip.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libaid_$name.so $objs tools/exp/${F%.hip}_$name.o
