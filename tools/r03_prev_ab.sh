#!/bin/bash
# same-box A/B of the current library against audio_inpainting_diffusion_amd/libaid_hip_prev.so (a build of an earlier commit, not tracked)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=audio_inpainting_diffusion_amd
cp $P/libaid_hip.so /tmp/new.so
run() { timeout 600 python bench.py --no-cpu-baseline --roof-steps 1 "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for which in new prev; do
  if [ $which = prev ]; then cp $P/libaid_hip_prev.so $P/libaid_hip.so; else cp /tmp/new.so $P/libaid_hip.so; fi
  echo -n "rep $rep $which batch 8: "; run
  echo -n "rep $rep $which batch 8 xi=0: "; run --xi 0
  echo -n "rep $rep $which batch 1: "; run --batch 1 --steps 4 --warmup 2
done; done
cp /tmp/new.so $P/libaid_hip.so
