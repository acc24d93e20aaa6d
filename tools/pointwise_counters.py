#!/usr/bin/env python
"""Join the launch manifest of tools/pointwise_probe.py with the per-dispatch FETCH_SIZE / WRITE_SIZE counters of two rocprofv3 --pmc passes.
Multi-kernel entry points (aid_group_stats = partial + final, aid_norm_bwd = coef + main, aid_group_dot = partial) are summed per entry point.
FETCH_SIZE on gfx950 counts 64 B per 128-B request of a wide streaming read (MI355X_MICROARCH.md, HBM section): the table prints the raw counter
and the x2-corrected value; WRITE_SIZE is printed as reported (KB units -> bytes), calibrated against `add2 (copy)`, whose writes are exactly 4 B per element.
usage: pointwise_counters.py manifest.json <dir of the FETCH pass> <dir of the WRITE pass>"""
import csv, glob, json, sys

man = json.load(open(sys.argv[1]))
OURS = ("norm_bwd", "scale_act", "group_stats", "group_dot", "add2_kernel", "resample_kernel")


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in OURS)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def group(rows):
    """consecutive dispatches that belong to one C-ABI call: helper kernels (…_final, …_coef, …partial) attach to their neighbour"""
    out, i = [], 0
    while i < len(rows):
        nm = rows[i]["Kernel_Name"]
        v = float(rows[i]["Counter_Value"])
        names = [nm.split("(")[0]]
        if "group_stats_partial" in nm and i + 1 < len(rows) and "group_stats_final" in rows[i + 1]["Kernel_Name"]:
            v += float(rows[i + 1]["Counter_Value"]); names.append("group_stats_final"); i += 1
        elif "norm_bwd_coef" in nm and i + 1 < len(rows) and "norm_bwd" in rows[i + 1]["Kernel_Name"]:
            v += float(rows[i + 1]["Counter_Value"]); names.append(rows[i + 1]["Kernel_Name"].split("(")[0]); i += 1
        out.append((names, v))
        i += 1
    return out


fe, wr = group(load(sys.argv[2])), group(load(sys.argv[3]))
print("\nPMC counters per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KB as reported -> MB):")
if len(fe) != len(man) or len(wr) != len(man):
    print("  dispatch count mismatch: manifest %d, fetch pass %d, write pass %d -- raw per-kernel sums follow" % (len(man), len(fe), len(wr)))
    for lab, rows in (("FETCH", fe), ("WRITE", wr)):
        agg = {}
        for names, v in rows:
            agg[names[-1]] = agg.get(names[-1], 0.0) + v
        print("  ", lab, {k: round(v / 1e3, 1) for k, v in agg.items()})
    sys.exit(0)
print("  %-28s %-18s %9s %11s %13s %10s %12s" % ("kernel", "shape", "alg MB", "FETCH MB", "FETCH x2 MB", "WRITE MB", "(Fx2+W)/alg"))
for m, (fn, f), (wn, w) in zip(man, fe, wr):
    fmb, wmb = f * 1024 / 1e6, w * 1024 / 1e6
    print("  %-28s C%-3d F%-3d T%-5d %9.1f %11.1f %13.1f %10.1f %12.2f" % (m["kernel"], m["C"], m["F"], m["T"], m["bytes"] / 1e6, fmb, 2 * fmb, wmb, (2 * fmb + wmb) / (m["bytes"] / 1e6)))
