#!/bin/bash
# round 6, GPU batch 10: 96-channel levels on the 2-D form by default (min-channels fix), T segments of 512: tests, probe, A/B
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py tests/test_gpu_vjp.py tests/test_gpu_network.py -x -q -m gpu -k "wino2d or norm_bwd or unet_full_cfgA_vs or fused_passes or full_size_guided" 2>&1 | grep -v amdgpu | tail -5 > $O/r06_b10_tests.txt
for b in 4; do for l in "L1 C96" "L2 C96"; do PROBE_ONLY="$l" PROBE_TF=8 timeout 300 python tools/w2d_probe.py layer $b | grep -v "^layer"; done; done 2>&1 | grep -v amdgpu > $O/r06_w2d_c96_tf8_probe3.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b10_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b10_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
run
run --w2d-c96-max-t 0
run
run --w2d-c96-max-t 0
run --batch 1
run --batch 1 --w2d-c96-max-t 0
run --batch 2
run --workload musicnet44k
run --workload musicnet44k --w2d-c96-max-t 0
timeout 600 python tools/plan_trace.py 8 2>&1 | grep -v amdgpu > $O/r06_trace_b8.txt
tail -3 $O/r06_b10_err.txt | grep -v amdgpu
