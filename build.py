#!/usr/bin/env python
"""Build libaid_hip.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.  No GPU needed to build."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "audio_inpainting_diffusion_amd")
SRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libaid_hip.so")


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(SRC, "*.h")) + [os.path.join(ROOT, "include", "aid_kernels.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(SRC, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(SRC, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in [s] + deps[len(srcs):]):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + SRC,
               "-Wno-unused-result", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
