#!/bin/bash
# usage: tools/c1_run.sh <AID_C1_CFG values...>   (1x1 probe shapes; 9 = old tiled kernel)
cd "$(dirname "$0")/.."
for shape in "8 64 192 128 1024 1 1 1" "8 96 192 192 512 1 1 1" "8 64 64 64 2048 1 1 1" "8 192 96 192 512 1 1 1" "8 128 512 384 64 1 1 1" "8 512 256 448 32 1 1 1" "8 256 96 256 256 1 1 1" "8 128 64 64 2048 1 1 1"; do
  for v in "$@"; do
    echo -n "c1cfg=$v  "; AID_C1_CFG=$v python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
