#!/bin/bash
# K-group instances of the F(4,3) kernel (deep levels at batch 1): parity tests, per-layer A/B (experiment library, AID_W4R_KS = 0 / 1), end to end
out=gpurun_out/r04_ks4_probe.txt
mkdir -p gpurun_out; : > $out
echo "== kernel tests (product library)" >> $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vjp.py -m gpu -q -x -k "wino or partial or epilogue or split" 2>&1 | tail -3 >> $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for b in 1 2; do echo "product batch=$b: $(timeout 600 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out; done
export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_ks.so
for ks in 0 1; do
  echo "== AID_W4R_KS=$ks (F(4,3) column: plain tiles / K-group instances)" >> $out
  AID_W4R_KS=$ks timeout 600 python tools/wino8_probe.py 1 2>&1 | grep -E "T32 |T64 |sum" | sed -e 's/| F(8,3).*//' >> $out
  echo "AID_W4R_KS=$ks batch=1: $(AID_W4R_KS=$ks timeout 600 python bench.py --batch 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done
cat $out
