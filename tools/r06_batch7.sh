#!/bin/bash
# round 6, GPU batch 7: deferred gradient copies after the key fix (parity + A/B); CU-masked sub-batch streams again with a non-null main stream
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_vjp.py tests/test_gpu_configs.py tests/test_gpu_network.py tests/test_gpu_dist.py tests/test_gpu_generic.py tests/test_gpu_spectral.py tests/test_gpu_training.py -x -q -m gpu -s -k "not eight_ranks and not two_ranks and not bench_self and not rccl" 2>&1 | grep -v "^configs\[1\] B=8 item\|amdgpu" | tail -30 > $O/r06_b7_tests.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b7_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_b7_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step', 'host_enqueue_ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac'], 'cu_split': j['config'].get('cu_split'), 'cu_partition': j['config'].get('cu_partition')}))
" >> $F; }
run
run --cu-split 16,16
run --cu-split 20,20
run --cu-split 32,32
run
run --cu-partition 96
run --cu-partition 64
run --batch 1
run --batch 1 --no-fold-copies
run --batch 1
run --batch 1 --no-fold-copies
run --batch 2
run --batch 2 --no-fold-copies
tail -3 $O/r06_b7_err.txt | grep -v amdgpu
