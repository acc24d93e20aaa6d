#!/bin/bash
# usage: tools/wino_exp_run.sh <variants...>   (runs on the GPU box)
cd "$(dirname "$0")/.."
for shape in "8 64 64 64 2048 5 3 2" "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 96 96 192 512 5 3 2" "8 256 256 384 64 5 3 8"; do
  for v in base "$@"; do
    if [ $v = base ]; then unset AID_LIB_PATH; else export AID_LIB_PATH=$PWD/tools/exp/libaid_exp$v.so; fi
    echo -n "exp=$v  "; PROBE_WINO=30 python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
