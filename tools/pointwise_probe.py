#!/usr/bin/env python
"""HBM-bound kernels of the path (statistics, activation / Winograd-transform pre-passes, normalisation backward, residual joins, FIR
resamplers, group dot) on the layer shapes of the 22 kHz network at batch 8: one launch sequence that can run

  * plain:            HIP-event timing, achieved TB/s against the ALGORITHMIC bytes of each kernel (printed table);
  * under rocprofv3:  `--manifest FILE` writes the launch order (kernel, shape, algorithmic bytes) so that tools/pointwise_counters.py can join
                      the per-dispatch FETCH_SIZE / WRITE_SIZE counters of two `--pmc` passes with it.
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

manifest = sys.argv[sys.argv.index("--manifest") + 1] if "--manifest" in sys.argv else None
reps = 1 if manifest else 20
B = 8
rows = []


def timeit(fn):
    if manifest:
        fn()
        torch.cuda.synchronize()
        return 0.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (C, F, T) in [(64, 64, 2048), (96, 128, 1024), (96, 192, 512), (128, 256, 256), (128, 320, 128), (256, 384, 64), (256, 448, 32)]:
    n = B * C * F * T
    x, gd, gy, out = (torch.randn(B, C, F, T, device="cuda") for _ in range(4))
    half = torch.randn(B, C, F, T // 2, device="cuda")
    sc = torch.rand(B, C, device="cuda") + 0.5
    stats = torch.rand(B, 8, 2, device="cuda")
    ws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT * 2 + 2), device="cuda", dtype=torch.float64)
    xv = torch.empty(B, C, F, 6 * (T // 4), device="cuda")
    scale = torch.empty(B, C, device="cuda"); gamma = torch.ones(C, device="cuda")
    cases = []
    p = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    cases.append(("norm_bwd", "aid_norm_bwd", p, 16 * n))
    pw = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    pw.wout, pw.wscale, pw.wscale_ld = L.view4(xv), sc.data_ptr(), sc.stride(0)
    cases.append(("norm_bwd_wino", "aid_norm_bwd", pw, 22 * n))
    cases.append(("scale_act_wino", "aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv), sc.data_ptr(), sc.stride(0), B, C, F, T, 1, 1), 10 * n))
    xv8 = torch.empty(B, C, F, 10 * (T // 8), device="cuda")                      # F(8,3) forms (round 4): 1.25x instead of 1.5x Winograd-domain tensor
    pw8 = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    pw8.wout, pw8.wscale, pw8.wscale_ld, pw8.wform = L.view4(xv8), sc.data_ptr(), sc.stride(0), 2
    cases.append(("norm_bwd_wino8", "aid_norm_bwd", pw8, 21 * n))
    cases.append(("scale_act_wino8", "aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv8), sc.data_ptr(), sc.stride(0), B, C, F, T, 1, 2), 9 * n))
    cases.append(("scale_act", "aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(out), sc.data_ptr(), sc.stride(0), B, C, F, T, 1, 0), 8 * n))
    cases.append(("group_stats", "aid_group_stats", L.GroupStatsParams(L.view4(x), B, C, F, T, 8, gamma.data_ptr(), None, 0, 1e-7, scale.data_ptr(), stats.data_ptr(), ws.data_ptr(), 0), 4 * n))
    cases.append(("group_dot", "aid_group_dot", L.GroupDotParams(L.view4(gd), L.view4(x), B, C, F, T, 8, ws.data_ptr()), 8 * n))
    cases.append(("add2 (2 in)", "aid_add2", L.Add2Params(L.view4(x), L.view4(gd), L.view4(out), B, C, F, T, 0.7, 0.7), 12 * n))
    cases.append(("add2 (copy)", "aid_add2", L.Add2Params(L.view4(x), L.view4(None), L.view4(out), B, C, F, T, 0.7, 0.0), 8 * n))
    cases.append(("resample down", "aid_resample", L.ResampleParams(L.view4(x), L.view4(half), B, C, F, T, 0, 0, 0), 6 * n))
    cases.append(("resample up", "aid_resample", L.ResampleParams(L.view4(half), L.view4(out), B, C, F, T // 2, 1, 0, 0), 6 * n))
    # (aid_kernels.h: T = length of the tensor passed as x; for adjoint = 1 that is the incoming gradient)
    cases.append(("resample down adjoint (+=)", "aid_resample", L.ResampleParams(L.view4(half), L.view4(out), B, C, F, T // 2, 0, 1, 1), 10 * n))
    cases.append(("resample up adjoint", "aid_resample", L.ResampleParams(L.view4(out), L.view4(half), B, C, F, T, 1, 1, 0), 6 * n))
    line = []
    for label, fn, params, nbytes in cases:
        if os.environ.get("PW_VERBOSE"):
            print("C%d F%d T%d %s" % (C, F, T, label), file=sys.stderr, flush=True)
        ms = timeit(lambda: L.call(fn, params))
        rows.append(dict(kernel=label, C=C, F=F, T=T, bytes=nbytes))
        if not manifest:
            line.append("%s %.0f us %.2f TB/s" % (label, ms * 1e3, nbytes / ms / 1e9))
    if not manifest:
        print("C%d F%d T%d (%.0f MB per tensor): " % (C, F, T, 4 * n / 1e6) + " | ".join(line))
if manifest:
    json.dump(rows, open(manifest, "w"))
