#!/bin/bash
# fused finalisation of the epilogue partials (aid_conv2d fin_mode): tests, then end-to-end A/B (--no-fin) at batch 1 / 2 / 8
out=gpurun_out/r04_fin_ab.txt
mkdir -p gpurun_out; : > $out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vjp.py -m gpu -q -x -k "epilogue" 2>&1 | tail -15 >> $out
timeout 1500 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_network.py -m gpu -q -x 2>&1 | tail -5 >> $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for b in 1 2 0; do
  if [ $b = 0 ]; then a=""; else a="--batch $b"; fi
  for f in "" "--no-fin" "" "--no-fin"; do
    echo "batch=$b $f: $(timeout 900 python bench.py $a $f --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
  done
done
cat $out
