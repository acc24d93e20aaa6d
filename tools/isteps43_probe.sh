#!/bin/bash
# the same issue schedule (first three kh steps) for the two-buffer F(4,3) instances (Cin <= 128 layers: three workgroups per CU): experiment library tools/exp/libaid_n2.so
out=gpurun_out/r04_isteps43_probe.txt; : > $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for v in 0 1; do
  if [ $v = 0 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_n2.so; fi
  echo "== variant $v" >> $out
  timeout 600 python tools/wino8_probe.py 1 4 2>&1 | grep -E "sum" | cut -c1-70 >> $out
done
for i in 1 2; do for v in 0 1; do
  if [ $v = 0 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_n2.so; fi
  echo "variant $v batch 8: $(timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
  echo "variant $v batch 1: $(timeout 900 python bench.py --batch 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done; done
cat $out
