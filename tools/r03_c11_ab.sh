#!/bin/bash
# A/B of the register-streamed 1x1 kernel (AID_C11_RS) end to end
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --no-cpu-baseline --roof-steps 1 "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for rs in 1 0; do
  echo -n "rep $rep batch 8 rs=$rs: "; AID_C11_RS=$rs run
  echo -n "rep $rep batch 8 xi=0 rs=$rs: "; AID_C11_RS=$rs run --xi 0
  echo -n "rep $rep batch 1 rs=$rs: "; AID_C11_RS=$rs run --batch 1 --steps 4 --warmup 2
  echo -n "rep $rep batch 3 rs=$rs: "; AID_C11_RS=$rs run --batch 3 --steps 4 --warmup 2
done; done
