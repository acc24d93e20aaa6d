#!/bin/bash
# round 6, GPU batch 18: the direct-to-LDS 1x1 kernel with K chunks of 32 (two 48-KB buffers: ONE eight-wave workgroup per CU instead of two) on its 128 x 256 instance
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
X() { if [ "$1" = product ]; then echo env; else echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; fi; }
for l in c11k32b; do echo "=== $l"; $(X $l) timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -2; done > $O/r06_b18_tests.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b18_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b18_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'fams': {k[:44]: [v['launches'], v['avg_launch_us'], v['frac_of_fp32_mfma_peak']] for k, v in r['families'].items() if 'conv11' in k}}))
" >> $F; }
for rep in 1 2 3; do for l in product c11k32b c11k32a; do run $(X $l) $B; done; done
grep -v "amdgpu\|AID_EXPERIMENT" $O/r06_b18_err.txt | tail -3
