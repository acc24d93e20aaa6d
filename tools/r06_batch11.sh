#!/bin/bash
# round 6, GPU batch 11: row-axis output transform folded into the GEMM of the HBM-bound 2-D launches (M halves): tests, per-layer probe, A/B against a no-fold build
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py tests/test_gpu_vjp.py tests/test_gpu_network.py tests/test_gpu_configs.py -x -q -m gpu -k "wino2d or norm_bwd or unet_full_cfgA_vs or fused_passes or full_size_guided or eight_free_running or network_vjp or cfgB" 2>&1 | grep -v amdgpu | tail -6 > $O/r06_b11_tests.txt
( echo "=== product build (fold M for Cin <= 128 on the 80-plane form) ==="; timeout 600 python tools/w2d_tf_probe.py 4 1 2>&1 | grep -v amdgpu
  echo "=== no-fold build ==="; AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_nofold.so timeout 600 python tools/w2d_tf_probe.py 4 1 2>&1 | grep -v amdgpu ) > $O/r06_w2d_foldm_layer_ab.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b11_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b11_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
NF="env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_nofold.so"
run $B
run $NF $B
run $B
run $NF $B
run $B --batch 1
run $NF $B --batch 1
run $B --batch 2
run $NF $B --batch 2
run $B --workload musicnet44k
run $NF $B --workload musicnet44k
grep -v amdgpu $O/r06_b11_err.txt | tail -3
