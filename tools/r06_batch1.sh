#!/bin/bash
# round 6, GPU batch 1: new parity tests + the CU-mask (spatial partition) experiment
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "cu_partitioned" -s 2>&1 | tail -15 > $O/r06_test_partition.txt
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "eight_free_running" -s 2>&1 | tail -60 > $O/r06_test_traj.txt
timeout 600 python tools/cu_mask_probe.py 4 256 384 64 2 > $O/r06_cu_mask_probe.txt 2>&1
timeout 300 python tools/cu_mask_probe.py 4 128 320 128 4 2>&1 | grep -A20 "== 3" >> $O/r06_cu_mask_probe.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
run() { echo "### $*" >> $O/r06_cu_partition_ab.txt; timeout 600 $B "$@" 2>>$O/r06_cu_partition_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'streams': j['config']['sub_batch_streams'], 'cu_partition': j['config'].get('cu_partition'), 'single_stream_ms': j['roofline']['all_conv']['single_stream_ms_per_step']}))
" >> $O/r06_cu_partition_ab.txt; }
run
run --cu-partition 64
run --cu-partition 96
run
run --cu-partition 48
run --cu-partition 32
run --streams 3 --cu-partition 64
run --streams 4 --cu-partition 64
run --streams 4 --cu-partition 96
run
tail -5 $O/r06_cu_partition_err.txt
