#!/bin/bash
# rocprofv3 kernel statistics of the default bench command (two sub-batch streams) and of the single-stream schedule.
# usage (on the GPU box): tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
for mode in streams2 streams1; do
  extra=""; [ $mode = streams1 ] && extra="--streams 1"
  rm -rf gpurun_out/prof_tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o p -- python bench.py --steps 3 --warmup 1 --roof-steps 1 --no-cpu-baseline $extra > gpurun_out/${tag}_bench_${mode}_under_rocprof.json 2> gpurun_out/${tag}_bench_${mode}_under_rocprof.err
  f=$(find gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${tag}_bench_${mode}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_bench_${mode}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("${mode}: sum of kernel durations %.1f ms over the whole process" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print("  %-64s calls %6s avg %9.1f us  %5.2f%%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
done
rm -rf gpurun_out/prof_tmp
