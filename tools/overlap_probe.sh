#!/bin/bash
# Do the two lanes of the launch plans actually overlap on the GPU?  rocprofv3 kernel trace of a B=1 bench run (HIP-graph replay) and of eager launches;
# reports, per mode, the wall span of the traced kernels, the sum of their durations and the time during which >= 2 kernels were running.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in graph eager; do
  rm -rf gpurun_out/ov_tmp
  extra=""; [ $mode = eager ] && extra="--no-graphs"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov_tmp -o p -- python bench.py --no-cpu-baseline --batch 1 --warmup 2 --steps 3 $extra > gpurun_out/ov_$mode.json 2>/dev/null
  python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/ov_tmp/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
# keep the last 40 % of the trace (steady state: timed steps)
t0 = ev[int(len(ev) * 0.6)][0]
ev = [e for e in ev if e[0] >= t0]
span = max(e[1] for e in ev) - ev[0][0]
busy = sum(e[1] - e[0] for e in ev)
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
cur = 0; last = pts[0][0]; ov = 0; idle = 0
for t, d in pts:
    if cur >= 2: ov += t - last
    if cur == 0: idle += t - last
    cur += d; last = t
qs = sorted({e[3] for e in ev}); ss = sorted({e[4] for e in ev})
d = json.load(open("gpurun_out/ov_$mode.json"))
print("$mode: %.2f evals/s; kernels %d, span %.1f ms, sum of durations %.1f ms, >=2 kernels running %.1f ms, GPU idle %.1f ms, queues %s streams %s" % (d["value"], len(ev), span / 1e6, busy / 1e6, ov / 1e6, idle / 1e6, qs, ss))
PY
done
rm -rf gpurun_out/ov_tmp
