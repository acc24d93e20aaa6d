#!/bin/bash
# rocprofv3 kernel statistics of the training-step bench (batch 4, 2 warm-up + 3 timed iterations) -> gpurun_out/<tag>_train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o p -- python tools/train_bench.py 4 3 > gpurun_out/${tag}_train_bench_under_rocprof.txt 2>&1
cp "$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_train_kernel_stats.csv
rm -rf gpurun_out/prof_tmp
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f ms (5 iterations)" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:12]:
    print("  %-64s calls %6s avg %9.1f us  %5.2f%%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
