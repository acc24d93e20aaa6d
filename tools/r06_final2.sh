#!/bin/bash
# round 6: trimmed GPU suite once more (timing), PMC counters of the 2-D form's kernels on the 80-plane form (incl. the folded GEMM)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=15 > $O/r06_gpu_suite_trimmed.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_trimmed.txt 2>&1
export PROBE_TF=8
{ tools/pmc_w2d.sh "L5 C256" 4; tools/pmc_w2d.sh "L3 C128" 4; tools/pmc_w2d.sh "L2 C96" 4; } > $O/r06_w2d_pmc.txt 2>&1
tail -3 $O/r06_gpu_suite_trimmed.txt
