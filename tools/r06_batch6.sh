#!/bin/bash
# round 6, GPU batch 6: deferred gradient copies (fold_grad_copies): parity + A/B; masked-stream end-to-end diagnostic
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_vjp.py tests/test_gpu_configs.py tests/test_gpu_network.py tests/test_gpu_dist.py tests/test_gpu_generic.py tests/test_gpu_spectral.py -x -q -m gpu -s -k "not eight_ranks and not two_ranks and not bench_self and not rccl" 2>&1 | grep -v "^configs\[1\] B=8 item\|amdgpu" | tail -30 > $O/r06_b6_tests.txt
timeout 600 python tools/cu_mask_e2e_probe.py 2>&1 | grep -v amdgpu > $O/r06_cu_mask_e2e_probe.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b6_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b6_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
run
run --no-fold-copies
run
run --no-fold-copies
run --batch 1
run --batch 1 --no-fold-copies
run --batch 1
run --batch 1 --no-fold-copies
run --batch 2
run --batch 2 --no-fold-copies
run --streams 3
tail -3 $O/r06_b6_err.txt | grep -v amdgpu
