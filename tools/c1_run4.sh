#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 64 192 128 1024 1 1 1" "8 96 192 192 512 1 1 1" "8 64 64 64 2048 1 1 1" "8 64 128 64 2048 1 1 1" "8 96 64 128 1024 1 1 1" "8 96 256 256 256 1 1 1"; do
  for v in 0 9; do
    echo -n "c1cfg=$v  "; AID_C1_CFG=$v python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
