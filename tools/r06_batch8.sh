#!/bin/bash
# round 6, GPU batch 8: the 96-channel levels on the 2-D form with 80 planes (96 x 128 GEMM tiles): tests, per-layer probe, end-to-end A/B
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu 2>&1 | tail -4 | grep -v amdgpu > $O/r06_b8_tests.txt
for b in 4 1; do PROBE_ONLY="L1 C96" PROBE_TF=8 timeout 300 python tools/w2d_probe.py layer $b; PROBE_ONLY="L2 C96" PROBE_TF=8 timeout 300 python tools/w2d_probe.py layer $b; PROBE_ONLY="L3d C96" PROBE_TF=8 timeout 300 python tools/w2d_probe.py layer $b; done 2>&1 | grep -v amdgpu > $O/r06_w2d_c96_tf8_probe.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b8_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b8_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
run
run --w2d-c96-max-t 256
run --w2d-c96-max-t 512
run --w2d-c96-max-t 1024
run
run --w2d-c96-max-t 256
run --w2d-c96-max-t 512
run --w2d-c96-max-t 1024
run --batch 1
run --batch 1 --w2d-c96-max-t 256
run --batch 1 --w2d-c96-max-t 512
run --batch 1 --w2d-c96-max-t 1024
run --batch 2
run --batch 2 --w2d-c96-max-t 1024
tail -3 $O/r06_b8_err.txt | grep -v amdgpu
