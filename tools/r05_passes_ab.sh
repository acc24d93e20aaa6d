#!/bin/bash
# A/B of the 2-D Winograd transform passes (round 5): first version (AID_W2D_LEGACY=3) against the LDS-staged input pass / T-transform-first output pass.
# (historical: AID_W2D_LEGACY existed only in the intermediate build that carried both versions of the passes; the first versions were removed after this A/B)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for B in 4 1; do
  for leg in 3 0; do
    echo "== AID_W2D_LEGACY=$leg  batch $B"
    AID_W2D_LEGACY=$leg timeout 600 python tools/w2d_probe.py layer $B 2>&1 | grep -v amdgpu.ids
  done
done > $O/r05_w2d_passes_ab.txt
for leg in 3 0 3 0; do
  echo "== AID_W2D_LEGACY=$leg"; AID_W2D_LEGACY=$leg python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-220
done > $O/r05_w2d_passes_bench_ab.txt
