#!/bin/bash
# Round-2 measurement sweep (run on the GPU box; everything lands in gpurun_out/r02_*; copy into profiles/ what should be judged).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null; }
{
  run --xi 0
  run --workload librispeech16k --gap-ms 25
  run --workload librispeech16k --gap-ms 50
  run --workload librispeech16k --gap-ms 100
  run --workload musicnet44k
  run --workload musicnet44k --xi 0
  run --task spectrogram
  run --batch 1 --warmup 2
  run --batch 2 --warmup 2
  run --batch 16 --steps 2
  run --streams 1
  run --streams 2
} > $O/r02_bench_variants.jsonl
# PMC: HBM traffic of the conv kernels (FETCH_SIZE and WRITE_SIZE in SEPARATE passes), single-stream schedule
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 1 > /dev/null 2>&1
done
python tools/conv_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r02_conv_traffic.json > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# PMC: matrix-pipe occupancy of the dominant kernel on its main shapes
for shape in "8 256 256 384 64 5 3 4" "8 256 256 448 32 5 3 4" "8 128 128 320 128 5 3 4" "8 64 64 64 2048 5 3 1"; do tools/pmc_conv.sh "$shape" 1; done > $O/r02_wino_pmc.txt 2>&1
tools/pmc_conv.sh "8 96 96 192 512 5 3 4" 1 >> $O/r02_wino_pmc.txt 2>&1
