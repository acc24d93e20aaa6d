#!/bin/bash
# EXPERIMENT: row-shared F(4,3) kernel variants on the main layer shapes: staging of raw taps vs the 30-tap pack, slot interleave, buffers / workgroups per CU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "0 0 0" "0 1 0" "0 0 1" "0 1 1" "1 1 1" "2 1 1" "3 1 1" "1 0 0"; do
  set -- $cfg
  AID_W4R_MODE=$1 AID_W4R_RAW=$2 AID_W4R_IL=$3 python tools/w4r_ab.py 20 2>&1 | grep -v amdgpu.ids
done
done
