#!/bin/bash
# when the direct-to-LDS loads of the next chunk are issued in the F(8,3) row-shared kernel: over all five kh steps of the current chunk (round-4 build up to here)
# or in its first 3 / 2 / 1 steps (experiment libraries tools/exp/libaid_i{3,2,1}.so); per-layer sums (tools/wino8_probe.py) and end to end
out=gpurun_out/r04_isteps_probe.txt; : > $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for v in 5 3 2 1; do
  if [ $v = 5 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_i$v.so; fi
  echo "== issue steps $v" >> $out
  timeout 600 python tools/wino8_probe.py 4 8 2>&1 | grep -E "sum" | cut -c1-70 >> $out
done
for i in 1 2; do for v in 5 3 2 1; do
  if [ $v = 5 ]; then unset AID_EXPERIMENT AID_LIB_PATH; else export AID_EXPERIMENT=1 AID_LIB_PATH=$PWD/tools/exp/libaid_i$v.so; fi
  echo "issue steps $v: $(timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done; done
cat $out
