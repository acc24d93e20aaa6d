#!/bin/bash
# Round-6 measurement sweep (run on the GPU box; everything lands in gpurun_out/r06_*; copy into profiles/ what should be judged).
# usage: tools/r06_measure.sh [part ...]   parts: bench variants small prof pmc probe pointwise traces e2e train c11   (default: all)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
parts=${@:-bench variants small prof pmc pointwise traces e2e train c11}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null; }
for part in $parts; do case $part in
bench)
  python bench.py 2>/dev/null > $O/r06_bench_guided.json                       # the driver's command (with the CPU baseline)
  run --streams 1 > $O/r06_bench_streams1.json
  run --steps 20 > $O/r06_bench_steps20.json ;;
variants)
  { run --xi 0
    run --workload librispeech16k --gap-ms 25
    run --workload librispeech16k --gap-ms 50
    run --workload librispeech16k --gap-ms 100
    run --workload musicnet44k
    run --workload musicnet44k --xi 0
    run --task spectrogram
    run --batch 16 --steps 2
    run --streams 3
    run --wino-forms 4,8,45
    run --wino-forms 4,8
    run --wino-forms 4
    run --no-fused-norm-bwd
    run --no-fold-copies
    run --no-pair-merge
    run --mfma-split 6
    run --mfma-split 6 --batch 1 --steps 4 --warmup 2
    run --mfma-split 6 --workload musicnet44k
  } > $O/r06_bench_variants.jsonl ;;
small)
  { for b in 1 2 3 4; do run --batch $b --steps 4 --warmup 2; done
    for b in 1 2 4; do run --batch $b --steps 4 --warmup 2 --wino-forms 4,8,45; done
    for b in 1 2; do run --batch $b --steps 4 --warmup 2 --no-fold-copies; done
    run --batch 1 --steps 4 --warmup 2 --no-lanes
    run --batch 1 --steps 4 --warmup 2 --graphs
    run --batch 2 --steps 4 --warmup 2 --no-lanes
  } > $O/r06_small_batch.jsonl ;;
prof)
  tools/profile_bench.sh r06 > $O/r06_profile_summary.txt 2>&1 ;;
timeline)
  rm -rf $O/prof_tl
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o p -- python bench.py --steps 3 --warmup 1 --roof-steps 1 --no-cpu-baseline > /dev/null 2>&1
  python tools/overlap_timeline.py $(find $O/prof_tl -name "*kernel_trace.csv" | head -1) > $O/r06_overlap_timeline.txt 2>&1
  rm -rf $O/prof_tl ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --roof-steps 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
  done
  python tools/conv_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r06_conv_traffic.json > /dev/null
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
pmcshapes)
  for shape in "8 256 256 384 64 5 3 4" "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 64 64 64 2048 5 3 1" "8 96 96 192 512 5 3 4"; do tools/pmc_conv.sh "$shape" 2; tools/pmc_conv.sh "$shape" 1; done > $O/r06_wino_pmc.txt 2>&1 ;;
probe)
  python tools/w2d_probe.py layer 4 > $O/r06_w2d_layer_probe.txt 2>&1 ;;
pointwise)
  tools/pointwise_counters.sh r06 > /dev/null 2>&1 ;;
traces)
  python tools/plan_trace.py 8 maestro22k --list > $O/r06_trace_b8.txt 2>/dev/null
  python tools/plan_trace.py 1 maestro22k --list > $O/r06_trace_b1.txt 2>/dev/null
  python tools/plan_trace.py 4 musicnet44k > $O/r06_trace_cfgB_b4.txt 2>/dev/null
  python bench.py --no-cpu-baseline --streams 1 --conv-table > /dev/null 2> $O/r06_conv_table.txt ;;
streams)
  { for rep in 1 2 3; do for st in 2 3; do echo "== rep $rep --streams $st"; run --streams $st | cut -c1-200; done; done; } > $O/r06_streams_2_vs_3.txt ;;
e2e)
  { python tools/e2e_run.py maestro22k 8; python tools/e2e_run.py maestro22k 1; python tools/e2e_run.py musicnet44k 4; python tools/e2e_run.py librispeech16k 16; } > $O/r06_e2e_full_runs.txt 2>/dev/null ;;
c11)
  { for rs in 0 1; do echo "== AID_C11_RS=$rs (1: conv11_rs_kernel where it is selected; 0: the tile kernels)"; AID_C11_RS=$rs timeout 300 python tools/c11_probe.py 8; done
    # (the ablation block of profiles/r06_c11_probe.txt came from an experimental build with an AID_C11_MODE switch; the product kernel has none)
    for rs in 0 1; do echo "== batch 1, AID_C11_RS=$rs"; AID_C11_RS=$rs timeout 300 python tools/c11_probe.py 1; done
  } 2>&1 | grep -v amdgpu.ids > $O/r06_c11_probe.txt ;;
suite)
  timeout 2400 python -m pytest tests -m gpu -q -rs --durations=15 > $O/r06_gpu_suite.txt 2>&1
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite.txt 2>&1 ;;
train)
  { python tools/train_bench.py 4 3; python tools/train_bench.py 8 3; } > $O/r06_train_bench.txt 2>/dev/null ;;
esac; done
ls -la $O | grep r06_
