#!/bin/bash
# round 6, GPU batch 14: folded-GEMM unit test + eight-rank self-launch test after the bench fix; 96-channel fold on the 128-wide four-wave kernel (experiment build) against the product
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py tests/test_gpu_dist.py -x -q -m gpu -s -k "folded_gemm or self_launches_eight" 2>&1 | grep -v amdgpu | tail -6 > $O/r06_b14_tests.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b14_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b14_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac'], 'fams': {k[:40]: [v['avg_launch_us'], v['frac_of_fp32_mfma_peak']] for k, v in r['families'].items() if 'fold' in k}}))
" >> $F; }
X() { echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; }
for rep in 1 2 3; do
run $B
run $(X c96wide) $B
done
run $B --batch 4
run $(X c96wide) $B --batch 4
run $B --workload musicnet44k
run $(X c96wide) $B --workload musicnet44k
grep -v "amdgpu\|AID_EXPERIMENT" $O/r06_b14_err.txt | tail -3
