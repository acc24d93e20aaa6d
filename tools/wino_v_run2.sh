#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 128 128 256 256 5 3 16" "8 256 256 384 64 5 3 8"; do
  for v in "$@"; do
    echo -n "VCFG=$v  "; AID_WINO_VCFG=$v PROBE_V=1 PROBE_WINO=30 python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
