#!/bin/bash
# Samples GPU clock / power / temperature (rocm-smi) every 0.5 s while the default bench runs: is the step clock- or power-limited?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for i in $(seq 1 140); do rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | head -1; sleep 0.5; done ) > gpurun_out/r05_clock_power_samples.csv &
S=$!
python bench.py --no-cpu-baseline --steps 20 --warmup 2 2>/dev/null | cut -c1-160
wait $S
rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | head -1 > gpurun_out/r05_clock_power_header.csv
