#!/bin/bash
# A/B of the pair instance and the batch-1 split-K instances (AID_W4R_SPLIT = tile-count threshold) end to end.
# AID_W4R_PAIR was a switch of the build this was run on (profiles/r03_w4r_ab.txt); the two-launch path has since been removed, so pair=0 now equals pair=1.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --no-cpu-baseline --roof-steps 1 "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for b in 1 2 3; do
  for cfg in "1 0" "1 230" "1 336" "0 0"; do set -- $cfg
    echo -n "rep $rep batch $b pair=$1 split=$2: "; AID_W4R_PAIR=$1 AID_W4R_SPLIT=$2 run --batch $b --steps 4 --warmup 2
  done
done
for cfg in "1 0" "0 0"; do set -- $cfg
  echo -n "rep $rep batch 8 pair=$1: "; AID_W4R_PAIR=$1 AID_W4R_SPLIT=$2 run
done
done
