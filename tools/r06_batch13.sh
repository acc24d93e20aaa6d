#!/bin/bash
# round 6, GPU batch 13: folded-GEMM variants: three workgroups per CU, fold on the K = 256 levels too, workgroup threshold 512 -- against the product (768, 2 per CU, Cin <= 128)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b13_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b13_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
X() { echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; }
for rep in 1 2; do
run $B
run $(X wpc3) $B
run $(X cin256) $B
run $(X f512) $B
done
run $B --batch 2
run $(X f512) $B --batch 2
run $(X cin256) $B --batch 2
run $B --batch 4
run $(X f512) $B --batch 4
grep -v amdgpu $O/r06_b13_err.txt | grep -v "AID_EXPERIMENT" | tail -3
