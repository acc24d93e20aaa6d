#!/bin/bash
# usage: tools/wino_env_run.sh VAR v1 v2 ...   (runs the probe shapes with VAR=v on the GPU box)
cd "$(dirname "$0")/.."
var=$1; shift
for shape in "8 64 64 64 2048 5 3 2" "8 256 256 448 32 5 3 4" "8 256 256 448 32 5 3 64" "8 128 128 320 128 5 3 4" "8 128 128 256 256 5 3 16" "8 96 96 192 512 5 3 2" "8 96 96 128 1024 5 3 2" "8 256 256 384 64 5 3 8"; do
  for v in "$@"; do
    echo -n "$var=$v  "; env $var=$v PROBE_WINO=30 python tools/conv_probe.py $shape 20 -1 2>&1 | tail -1
  done
done
