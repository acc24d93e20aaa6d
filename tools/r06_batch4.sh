#!/bin/bash
# round 6, GPU batch 4: TF8 passes after the input-pass fix, GP = 2 output-pass variant, trace, launch cost on masked streams
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu 2>&1 | tail -3 > $O/r06_b4_tests.txt
( echo "=== product build (GP8 = 1) ==="; timeout 600 python tools/w2d_tf_probe.py 4 8 2>&1 | grep -v amdgpu
  echo "=== experiment build: two groups per thread in the F(8,3) output pass where a row has >= 16 groups ==="; AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_gp2.so timeout 600 python tools/w2d_tf_probe.py 4 8 2>&1 | grep -v amdgpu ) > $O/r06_w2d_tf8_layer_ab2.txt
timeout 300 python tools/cu_mask_probe.py --launch-cost 2>&1 | grep -v amdgpu > $O/r06_cu_mask_launch_cost.txt
timeout 600 python tools/plan_trace.py 8 2>&1 | grep -v amdgpu > $O/r06_trace_b8.txt
timeout 600 python tools/plan_trace.py 1 2>&1 | grep -v amdgpu > $O/r06_trace_b1.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_tf8_bench_ab2.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b4_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac'], 'avg_us': r['avg_launch_us'], 'step_executed_frac': r['step_executed_frac'], 'gemm': {k: r['kernels'].get('w2d_gemm_kernel', {}).get(k) for k in ('frac_of_fp32_mfma_peak', 'avg_launch_us', 'time_ms')}, 'wino8r_ms': r['kernels'].get('conv53_wino8r_kernel', {}).get('time_ms')}))
" >> $F; }
run --wino-forms 4,8,45
run
run --wino-forms 4,8,45
run
run --xi 0
run --workload librispeech16k
tail -3 $O/r06_b4_err.txt | grep -v amdgpu
