#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 64 64 64 2048 5 3 2" "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 256 256 384 64 5 3 8"; do
  for v in base "$@"; do
    if [ $v = base ]; then unset AID_LIB_PATH; else export AID_LIB_PATH=$PWD/tools/exp/libaid_exp$v.so; fi
    echo -n "exp=$v  "; PROBE_V=1 PROBE_WINO=30 python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
