#!/bin/bash
# round 6, GPU batch 12: which launches fold the row transform into the GEMM -- workgroup-count threshold 768 / 1024 (product) / 1900 against no fold
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_b12_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 "$@" 2>>$O/r06_b12_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac']}))
" >> $F; }
X() { echo "env AID_EXPERIMENT=1 AID_LIB_PATH=tools/exp/libaid_$1.so"; }
for rep in 1 2; do
run $(X nofold) $B
run $(X fold768) $B
run $B
run $(X fold1900) $B
done
for b in 2 1; do
run $(X nofold) $B --batch $b
run $(X fold768) $B --batch $b
run $B --batch $b
done
run $(X nofold) $B --workload musicnet44k
run $(X fold768) $B --workload musicnet44k
run $B --workload musicnet44k
run $(X nofold) $B --workload librispeech16k
run $B --workload librispeech16k
grep -v amdgpu $O/r06_b12_err.txt | grep -v "AID_EXPERIMENT" | tail -3
