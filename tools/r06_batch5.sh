#!/bin/bash
# round 6, GPU batch 5: norm_bwd folded into the 2-D input pass + JB rule; launch cost on masked streams; traces; end-to-end A/Bs
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py tests/test_gpu_vjp.py tests/test_gpu_configs.py tests/test_gpu_network.py -x -q -m gpu -s -k "wino2d or norm_bwd or fused_passes or full_size_guided or eight_free_running or unet_full_cfgA_vs or guided_whole_trajectory or network_vjp" 2>&1 | grep -v "^configs\[1\] B=8 item\|amdgpu" | tail -25 > $O/r06_b5_tests.txt
timeout 300 python tools/cu_mask_probe.py --launch-cost 2>&1 | grep -v amdgpu > $O/r06_cu_mask_launch_cost.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b5_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b5_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac'], 'avg_us': r['avg_launch_us'], 'step_executed_frac': r['step_executed_frac'], 'gemm': {k: r['kernels'].get('w2d_gemm_kernel', {}).get(k) for k in ('frac_of_fp32_mfma_peak', 'avg_launch_us', 'time_ms')}, 'wino8r_ms': r['kernels'].get('conv53_wino8r_kernel', {}).get('time_ms')}))
" >> $F; }
run
run --no-fused-norm-bwd
run --wino-forms 4,8,45
run
run --no-fused-norm-bwd
run --wino-forms 4,8,45
run --batch 1
run --batch 1 --no-fused-norm-bwd
run --batch 1 --wino-forms 4,8,45
run --batch 2
run --xi 0
run --workload librispeech16k
run --workload musicnet44k
timeout 600 python tools/plan_trace.py 8 2>&1 | grep -v amdgpu > $O/r06_trace_b8.txt
timeout 600 python tools/plan_trace.py 1 2>&1 | grep -v amdgpu > $O/r06_trace_b1.txt
tail -3 $O/r06_b5_err.txt | grep -v amdgpu
