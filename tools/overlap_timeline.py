#!/usr/bin/env python
"""Who runs beside whom under the two-stream schedule?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --steps 3 --warmup 1 --roof-steps 1` and, over the
timed region (the longest stretch in which kernels of TWO streams interleave), reports the share of wall time with 0 / 1 / 2+ kernels in flight and, for the
time with two, which classes are paired: M = MFMA-bound (conv53_wino*, w2d_gemm, conv11_*, conv_mfma, attention), H = HBM-bound (everything else).
   python tools/overlap_timeline.py <kernel_trace.csv>"""
import csv, sys, collections

MFMA = ("conv53_wino", "w2d_gemm", "conv11_", "conv_mfma", "time_attention", "attn_bwd", "gemm_skinny")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M" if any(m in name for m in MFMA) else "H", r.get("Stream_Id", r.get("Queue_Id", "0")), name))
rows.sort()
t_first, t_last = rows[0][0], max(r[1] for r in rows)
# the two-stream part of the run: queues other than the busiest one appear only there; take the span of the second-busiest queue's kernels
byq = collections.Counter(r[3] for r in rows)
qs = [q for q, _ in byq.most_common(3)]
if len(qs) >= 2:
    second = [r for r in rows if r[3] == qs[1]]
    lo, hi = second[0][0], max(r[1] for r in second)
else:
    lo, hi = t_first, t_last
ev = []
for s, e, c, q, n in rows:
    if e <= lo or s >= hi:
        continue
    ev.append((max(s, lo), 1, c)); ev.append((min(e, hi), -1, c))
ev.sort()
cur = collections.Counter(); last = lo
dur = collections.Counter()
for t, d, c in ev:
    key = "".join(sorted("M" * cur["M"] + "H" * cur["H"]))
    dur[key if len(key) <= 2 else key[:2] + "+"] += t - last
    cur[c] += d; last = t
tot = hi - lo
print(f"two-stream span {tot / 1e6:.1f} ms, {sum(1 for r in rows if lo <= r[0] < hi)} kernels on queues {dict(byq.most_common(4))}")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1]):
    print(f"  in flight {k or 'none':6s} {100 * v / tot:5.1f} %")
