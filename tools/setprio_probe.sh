#!/bin/bash
# s_setprio in the row-shared Winograd kernels: product (none) / p1 (priority 1 around every 10-MFMA cluster) / p2 (priority 1 for the whole K loop, 0 in
# prologue and epilogue) -- experiment libraries tools/exp/libaid_p{1,2}.so; per-layer sums (tools/wino8_probe.py) and end to end
out=gpurun_out/r04_setprio_probe.txt; : > $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for v in 0 1 2; do
  if [ $v = 0 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_p$v.so; fi
  echo "== variant $v" >> $out
  timeout 600 python tools/wino8_probe.py 4 8 2>&1 | grep -E "sum" | cut -c1-70 >> $out
done
for i in 1 2; do for v in 0 1 2; do
  if [ $v = 0 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_p$v.so; fi
  echo "variant $v: $(timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done; done
cat $out
