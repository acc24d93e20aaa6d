#!/bin/bash
# HBM traffic of every HBM-bound kernel from the PMC counters (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, MI355X_MICROARCH.md) next to its
# algorithmic bytes and its event-timed bandwidth: gpurun_out/<tag>_pointwise_bandwidth.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03}
out=gpurun_out/${tag}_pointwise_bandwidth.txt
timeout 300 python tools/pointwise_probe.py 2>/dev/null > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pw_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pw_$c -o p -- python tools/pointwise_probe.py --manifest gpurun_out/pw_manifest.json > /dev/null 2>&1
done
python tools/pointwise_counters.py gpurun_out/pw_manifest.json gpurun_out/pw_FETCH_SIZE gpurun_out/pw_WRITE_SIZE >> $out
rm -rf gpurun_out/pw_FETCH_SIZE gpurun_out/pw_WRITE_SIZE gpurun_out/pw_manifest.json
cat $out
