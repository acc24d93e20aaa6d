#!/bin/bash
# end-to-end A/B of the K-group F(8,3) instances: product library (rule: tiles <= 256) against the experiment library with AID_W8R_KS = 0 (never) / 2 (extended rule)
out=gpurun_out/r04_ks_bench.txt
mkdir -p gpurun_out; : > $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', d['roofline']['frac'])"; }
for b in 1 2 3; do
  echo "product batch=$b: $(timeout 600 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done
echo "product batch=6 (headline config): $(timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_ks.so
for ks in 0 2; do
  for b in 1 2 3 0; do
    if [ $b = 0 ]; then a=""; else a="--batch $b"; fi
    echo "AID_W8R_KS=$ks batch=$b: $(AID_W8R_KS=$ks timeout 900 python bench.py $a --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
  done
done
cat $out
