#!/usr/bin/env python
"""Average HBM bytes per aid_conv2d launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE
runs, as MI355X_MICROARCH.md prescribes).   usage: conv_traffic.py <dir with fetch pass> <dir with write pass> <out.json>"""
import collections, csv, glob, hashlib, json, os, sys


def kernel_sources_sha16():
    """hash of the kernel sources (csrc/*.hip, *.h, include/aid_kernels.h): bench.py reports this profile's traffic only for the same build"""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "audio_inpainting_diffusion_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "audio_inpainting_diffusion_amd", "csrc", "*.h"))
                    + [os.path.join(root, "include", "aid_kernels.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


CONV = ("w2d_gemm_kernel", "w2d_output_kernel", "w2d_input_kernel", "conv53_wino8r_kernel", "conv53_wino4r_kernel", "conv53_wino4v_kernel", "conv53_wino4_kernel", "conv_mfma_kernel", "conv1x1_stream_kernel", "conv11_dma_kernel", "conv11_rs_kernel", "conv_small_cout_kernel",
        "conv_small_cin_kernel", "conv53_dma_kernel")

def avg(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    tot, n = 0.0, 0
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"]
        name = next((c for c in CONV if c in k), None)
        if name is None: continue
        v = float(r["Counter_Value"])
        tot += v; n += 1
        per[name][0] += v; per[name][1] += 1
    return tot / max(1, n), n, {k: [v[0] / v[1], v[1]] for k, v in per.items()}

fetch, nf, pf = avg(sys.argv[1], "FETCH_SIZE")
write, nw, pw = avg(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 1",
       "kernel_sources_sha16": kernel_sources_sha16(),
       "kernel": "all aid_conv2d kernels (" + ", ".join(sorted(pf)) + ")", "launches": nf,
       "FETCH_SIZE_kb_avg_per_launch": round(fetch, 1), "WRITE_SIZE_kb_avg_per_launch": round(write, 1),
       "per_kernel_FETCH_SIZE_kb_avg": {k: round(v[0], 1) for k, v in pf.items()}, "per_kernel_WRITE_SIZE_kb_avg": {k: round(v[0], 1) for k, v in pw.items()},
       "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected",
       "per_kernel_hbm_bytes_corrected": {k: int((2 * pf[k][0] + pw.get(k, [0.0])[0]) * 1024) for k in pf},
       "hbm_bytes_per_conv_launch_corrected": int((2 * fetch + write) * 1024), "hbm_bytes_per_conv_launch_raw": int((fetch + write) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
