#!/bin/bash
# A/B of the K-group F(8,3) instances (experiment library tools/exp/libaid_ks.so, AID_W8R_KS = 0 / 1) per layer and batch; writes gpurun_out/r04_ks_probe.txt
export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_ks.so
out=gpurun_out/r04_ks_probe.txt
mkdir -p gpurun_out
: > $out
echo "== parity tests with the K-group instances forced" >> $out
AID_W8R_KS=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vjp.py -m gpu -q -x -k "winograd8 or partial or norm_bwd" 2>&1 | tail -5 >> $out
for ks in 0 1; do
  echo "== AID_W8R_KS=$ks" >> $out
  AID_W8R_KS=$ks timeout 900 python tools/wino8_probe.py 1 2 3 2>&1 | sed -e 's/| stream-K.*| library/| library/' >> $out
done
