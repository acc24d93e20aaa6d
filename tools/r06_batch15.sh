#!/bin/bash
# round 6, GPU batch 15: the folded GEMM with a deeper / wider prefetch (NBUF 4 / 5, K chunks of 32) and the 96-channel panels on three waves (exact 96 x 64 tiles, K chunks of 24)
# against the product (four-wave 128 x 64 instance for both, K chunks of 16, three buffers): unit test of the folded path, per-layer times, end to end
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
X() { if [ "$1" = product ]; then echo env; else echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; fi; }
LIBS="product nb4 nb5 kc32 w3 w3nb4"
for l in $LIBS; do echo "=== $l"; $(X $l) timeout 300 python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu -s -k "folded_gemm" 2>&1 | grep -v amdgpu | grep "folded GEMM\|passed\|failed\|Error" ; done > $O/r06_b15_tests.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b15_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b15_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'frac': r['frac'], 'fams': {k[:44]: [v['avg_launch_us'], v['frac_of_fp32_mfma_peak']] for k, v in r['families'].items() if 'fold' in k}}))
" >> $F; }
for rep in 1 2; do for l in $LIBS; do run $(X $l) $B; done; done
for l in product nb5 kc32 w3; do echo "=== $l"; $(X $l) timeout 400 python tools/w2d_tf_probe.py 4 2>&1 | grep -v amdgpu | grep "^L[1234]"; done > $O/r06_b15_layer_ab.txt
for l in product nb5 kc32; do run $(X $l) $B --batch 4; done
grep -v "amdgpu\|AID_EXPERIMENT" $O/r06_b15_err.txt | tail -3
