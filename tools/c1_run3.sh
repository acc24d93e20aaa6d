#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "8 128 64 64 2048 1 1 1" "8 192 96 192 512 1 1 1" "8 128 512 384 64 1 1 1" "8 512 256 448 32 1 1 1" "8 256 96 256 256 1 1 1" "8 256 128 320 128 1 1 1" "8 192 64 128 1024 1 1 1" "8 128 256 320 128 1 1 1"; do
  for v in 0 1; do
    echo -n "dma=$v  "; AID_C1_DMA=$v python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
