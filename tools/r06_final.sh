#!/bin/bash
# round 6: the driver's sequence once more on the final tree -- GPU suite (with durations), smoke, the bench line with roofline.traffic filled from the committed PMC profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=15 > $O/r06_gpu_suite_final.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_final.txt 2>&1
python bench.py 2>/dev/null > $O/r06_bench_guided_final.json
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>$O/r06_bench_gpus2_shared.err | tail -1 | cut -c1-600 > $O/r06_bench_gpus2_shared.txt
tail -3 $O/r06_gpu_suite_final.txt
