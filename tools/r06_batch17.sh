#!/bin/bash
# round 6, GPU batch 17 (small batches): the unfolded 96-channel GEMM on three waves (96 x 128 / 96 x 64 tiles, K chunks of 24) against the two-wave 96 x 128 instance,
# and the fold from 384 / 256 workgroups with the re-cut instances -- batch 1 and 2
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
X() { if [ "$1" = product ]; then echo env; else echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; fi; }
for l in u3a u3c; do echo "=== $l"; $(X $l) timeout 600 python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -2; done > $O/r06_b17_tests.txt
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b17_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b17_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'frac': r['frac'], 'fams': {k[:44]: [v['launches'], v['avg_launch_us'], v['frac_of_fp32_mfma_peak']] for k, v in r['families'].items() if 'w2d_gemm' in k}}))
" >> $F; }
for rep in 1 2; do for l in product u3a u3c f384 f256 u3af384; do run $(X $l) $B --batch 1; done; done
for rep in 1 2; do for l in product u3a u3c f384 f256 u3af384; do run $(X $l) $B --batch 2; done; done
for l in product f384 f256; do run $(X $l) $B --batch 8 --steps 6; done
grep -v "amdgpu\|AID_EXPERIMENT" $O/r06_b17_err.txt | tail -3
