#!/bin/bash
# round 6, GPU batch 16: the folded GEMM's instances after batch 15 (three-wave 96 x 64 with K chunks of 24, four-wave 128 x 64 with chunks of 32) as the product,
# against the build before (both panels on the four-wave instance, chunks of 16), the three-wave instance with chunks of 48 / two buffers, and the fold from 512 workgroups
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
X() { if [ "$1" = product ]; then echo env; else echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; fi; }
timeout 900 python -m pytest tests/test_gpu_wino2d.py tests/test_gpu_network.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 > $O/r06_b16_tests.txt
timeout 300 python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu -s -k "folded_gemm" 2>&1 | grep "folded GEMM\|passed\|failed\|Error" >> $O/r06_b16_tests.txt
$(X w3k48) timeout 300 python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu -s -k "folded_gemm and case1" 2>&1 | grep "folded GEMM\|passed\|failed\|Error" >> $O/r06_b16_tests.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b16_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b16_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'frac': r['frac'], 'fams': {k[:44]: [v['launches'], v['avg_launch_us'], v['frac_of_fp32_mfma_peak']] for k, v in r['families'].items() if 'fold' in k}}))
" >> $F; }
for rep in 1 2 3; do for l in c96wide product w3k48 f512b; do run $(X $l) $B; done; done
for l in c96wide product f512b; do run $(X $l) $B --batch 4; done
for l in product f512b product f512b; do run $(X $l) $B --batch 2; done
for l in c96wide product; do run $(X $l) $B --workload musicnet44k; done
grep -v "amdgpu\|AID_EXPERIMENT" $O/r06_b16_err.txt | tail -3
