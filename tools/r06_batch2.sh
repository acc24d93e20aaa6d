#!/bin/bash
# round 6, GPU batch 2: CU-masked sub-batch streams WITHOUT events (cu_split) + the type-lane partition again with re-used events
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_cu_split_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_cu_split_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print(json.dumps({k: j[k] for k in ('value', 'ms_per_step', 'host_enqueue_ms_per_step')} | {'streams': j['config']['sub_batch_streams'], 'cu_partition': j['config'].get('cu_partition'), 'cu_split': j['config'].get('cu_split'), 'single_stream_ms': j['roofline']['all_conv']['single_stream_ms_per_step']}))
" >> $F; }
run
run --cu-split 16,16
run --cu-split 20,20
run --cu-split 24,24
run
run --cu-split 28,28
run --cu-split 20,12 --split 5,3
run --cu-split 12,12,12 --streams 3
run --cu-split 16,16,16 --streams 3
run --cu-partition 96
run
tail -3 $O/r06_cu_split_err.txt
