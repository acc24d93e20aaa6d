#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 64 64 64 2048 5 3 2" "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 128 128 256 256 5 3 16" "8 96 96 192 512 5 3 2" "8 96 96 128 1024 5 3 2" "8 256 256 384 64 5 3 8"; do
  for v in 0 1; do
    echo -n "V=$v  "; PROBE_V=$v PROBE_WINO=30 python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
