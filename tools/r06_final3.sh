#!/bin/bash
# round 6: the evidence set once more on the final kernel sources -- GPU suite + smoke, PMC traffic of the driver's command, the bench line (traffic from that profile),
# rocprofv3 kernel statistics, counters of the folded GEMM levels, small batches, the other configurations, plan traces, timeline, complete runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=15 > $O/r06_gpu_suite_final.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_final.txt 2>&1
tail -3 $O/r06_gpu_suite_final.txt
tools/r06_measure.sh pmc > /dev/null
cp $O/r06_conv_traffic.json profiles/r06_conv_traffic.json
tools/r06_measure.sh bench prof > /dev/null
cp $O/r06_bench_guided.json $O/r06_bench_guided_final.json
export PROBE_TF=8
{ tools/pmc_w2d.sh "L5 C256" 4; tools/pmc_w2d.sh "L3 C128" 4; tools/pmc_w2d.sh "L2 C96" 4; } > $O/r06_w2d_pmc.txt 2>&1
unset PROBE_TF
tools/r06_measure.sh small variants traces timeline e2e > /dev/null
cut -c1-400 $O/r06_bench_guided_final.json
