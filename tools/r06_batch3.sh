#!/bin/bash
# round 6, GPU batch 3: the 2-D Winograd form with F(8,3) along T (x_wino = 4): tests, per-layer probe, end-to-end A/B
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_wino2d.py -x -q -m gpu 2>&1 | tail -15 > $O/r06_tf8_tests.txt
python -m pytest tests/test_gpu_network.py tests/test_gpu_vjp.py tests/test_gpu_configs.py -x -q -m gpu -s -k "unet_full_cfgA_vs or full_size_guided or eight_free_running or fused_passes_match" 2>&1 | grep -v "^configs\[1\] B=8 item" | tail -30 >> $O/r06_tf8_tests.txt
timeout 900 python tools/w2d_tf_probe.py 4 8 1 > $O/r06_w2d_tf8_layer_ab.txt 2>&1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roof-steps 1"
F=$O/r06_tf8_bench_ab.txt
run() { echo "### $*" >> $F; timeout 600 $B "$@" 2>>$O/r06_tf8_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(json.dumps({k: j[k] for k in ('value', 'ms_per_step')} | {'single_stream_ms': r['all_conv']['single_stream_ms_per_step'], 'dom': r['kernel'], 'frac': r['frac'], 'avg_us': r['avg_launch_us'], 'step_executed_frac': r['step_executed_frac']}))
" >> $F; }
run --wino-forms 4,8,45
run
run --wino-forms 4,8,45
run
run --batch 1 --wino-forms 4,8,45
run --batch 1
run --batch 2 --wino-forms 4,8,45
run --batch 2
run --workload musicnet44k --wino-forms 4,8,45
run --workload musicnet44k
tail -3 $O/r06_tf8_err.txt
