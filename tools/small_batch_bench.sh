#!/bin/bash
# evals/s of the headline workload at small batches (the reference's testers run B = 1)
for b in 1 2 3; do python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('batch', j['config']['segments_per_gpu'], 'value', j['value'], 'evals/s  ms/step', j['ms_per_step'], 'streams', j['config']['sub_batch_streams'])"; done
