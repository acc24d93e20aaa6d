#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 16 192 128 1024 1 1 1" "8 192 32 128 1024 1 1 1" "8 16 64 128 1024 1 1 1" "8 64 192 128 1024 1 1 1"; do
  for v in "$@"; do
    echo -n "c1cfg=$v  "; AID_C1_CFG=$v python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
