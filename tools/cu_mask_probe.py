#!/usr/bin/env python
"""Spatial partition of the chip with CU-masked streams: what does each side get?
  1. WHERE a masked stream runs: (XCC, SE, CU) of every workgroup of a spinning grid -- tells the mask's bit order apart (streams.partition_masks)
  2. streaming-copy bandwidth on the last n CUs of every XCD, n = 32 .. 128 (what an HBM-bound pass can hope for there)
  3. the 2-D Winograd form's passes + aid_norm_bwd ALONE on the n-CU partition, its GEMM ALONE on the other 256 - n CUs, and BOTH TOGETHER
     (GEMM of one sub-batch || passes of the other: the schedule the partition is meant to produce) against the unmasked streams
   python tools/cu_mask_probe.py [B] [C F T dil]
"""
import ctypes as C
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L
from audio_inpainting_diffusion_amd.streams import cu_masked_stream, partition_masks

LAUNCH_COST = "--launch-cost" in sys.argv
if LAUNCH_COST:
    sys.argv.remove("--launch-cost")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Cc, F, T, dil = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 384, 64, 2)
dev = "cuda"
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "cu_probe.so"))
probe.cu_where.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
probe.cu_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]


def where(stream, label):
    out = torch.zeros(4096, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        probe.cu_where(out.data_ptr(), 4096, 200000, stream.cuda_stream)
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    xcc, cu, sh, se = (v >> 24) & 0xF, (v >> 8) & 0xF, (v >> 12) & 1, (v >> 13) & 7
    ids = sorted({(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)})
    per_xcc = {x: sum(1 for i in ids if i[0] == x) for x in sorted({i[0] for i in ids})}
    print(f"  {label}: {len(ids)} distinct (xcc, se, sh, cu) slots; per XCC {per_xcc}")
    return ids


def timeit(fn, streams, reps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(reps)
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def main():
    print("== 1. where do masked streams run ==")
    full = torch.cuda.Stream()
    where(full, "unmasked stream")
    for n in (64,):
        for inter in (True, False):
            mw, pw = partition_masks(n, interleaved=inter)
            where(cu_masked_stream(pw), f"pass partition n={n} ({'round-robin bit order' if inter else 'contiguous bit order'})")
            where(cu_masked_stream(mw), f"MFMA partition 256-{n} ({'round-robin' if inter else 'contiguous'})")
    print("== 2. streaming copy (256 MiB -> 256 MiB, non-temporal) on the pass partition ==")
    nbytes = 256 << 20
    x, y = torch.randn(nbytes // 4, device=dev), torch.empty(nbytes // 4, device=dev)
    for n in (0, 128, 96, 64, 48, 32):
        s = full if n == 0 else cu_masked_stream(partition_masks(n)[1])
        ncu = n or 256
        row = []
        for unroll, wgs in ((1, 8), (4, 8), (8, 4), (8, 8)):
            def run(reps):
                for _ in range(reps):
                    probe.cu_copy(x.data_ptr(), y.data_ptr(), nbytes, ncu * wgs, unroll, s.cuda_stream)
            run(2)
            ms = timeit(run, [s])
            row.append(f"unroll {unroll} x {wgs} WG/CU: {2 * nbytes / ms / 1e9:5.2f} TB/s")
        print(f"  {ncu:3d} CUs: " + " | ".join(row))
    print(f"== 3. the 2-D form on partitions: B{B} C{Cc} F{F} T{T} d{dil} ==")
    lib = L.lib()
    N = int(lib.aid_conv2d_wino2d_positions(B, F, T, dil))
    xx, res, _ = (torch.randn(B, Cc, F, T, device=dev) for _ in range(3))
    w = torch.randn(Cc, Cc, 5, 3, device=dev) / math.sqrt(Cc * 15)
    wp, w2 = L.pack_conv_weight(w), L.pack_conv_weight_wino2d(w)
    isc = torch.rand(B, Cc, device=dev) + 0.5
    V, M = torch.empty(48 * Cc * N, device=dev), torch.empty(48 * Cc * N, device=dev)
    V2, M2, y2 = torch.empty_like(V), torch.empty_like(M), torch.empty_like(xx)
    sp = L.ScaleActParams(L.view4(xx), L.View(V2.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, Cc, F, T, 1, 3, dil)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xx), L.View(V.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, Cc, F, T, 1, 3, dil))
    gp = L.Wino2dGemmParams(w2.data_ptr(), V.data_ptr(), M.data_ptr(), 48, Cc, Cc, wp.shape[1], wp.shape[2], N, 0)
    L.call("aid_wino2d_gemm", L.Wino2dGemmParams(w2.data_ptr(), V.data_ptr(), M2.data_ptr(), 48, Cc, Cc, wp.shape[1], wp.shape[2], N, 0))
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.View(V2.data_ptr(), 0, 0, 0), L.view4(y2), L.view4(res), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), w2.data_ptr(), 48, 3
    p.out_scale, p.out_scale_ld = isc.data_ptr(), isc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cc, Cc, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = 0.7, 1.0
    p.ws, p.ws_bytes = M2.data_ptr(), M2.numel() * 4
    gd, gy, out = (torch.randn(B, Cc, F, T, device=dev) for _ in range(3))
    stats = torch.rand(B, 8, 2, device=dev)
    ws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT * 2 + 2), device=dev, dtype=torch.float64)
    nb = L.NormBwdParams(L.view4(gd), L.view4(xx), L.view4(gy), L.view4(out), B, Cc, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)

    def passes(s):                                   # one layer's worth of passes of the OTHER sub-batch: input pass, output pass, norm_bwd
        h = s.cuda_stream
        assert lib.aid_scale_act(C.addressof(sp), h) == 0 and lib.aid_conv2d_wino2d_output(C.addressof(p), h) == 0 and lib.aid_norm_bwd(C.addressof(nb), h) == 0

    def gemm(s):
        assert lib.aid_wino2d_gemm(C.addressof(gp), s.cuda_stream) == 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    g0 = timeit(lambda r: [gemm(sa) for _ in range(r)], [sa])
    g0 = timeit(lambda r: [gemm(sa) for _ in range(r)], [sa])
    h0 = timeit(lambda r: [passes(sb) for _ in range(r)], [sb])
    h0 = timeit(lambda r: [passes(sb) for _ in range(r)], [sb])
    both0 = timeit(lambda r: [(gemm(sa), passes(sb)) for _ in range(r)], [sa, sb])
    print(f"  unmasked: GEMM {1e3 * g0:.0f} us, passes (input + output + norm_bwd) {1e3 * h0:.0f} us, serial {1e3 * (g0 + h0):.0f} us, two free streams together {1e3 * both0:.0f} us per layer pair")
    for n in (32, 48, 64, 96, 128):
        mw, pw = partition_masks(n)
        sm, sh = cu_masked_stream(mw), cu_masked_stream(pw)
        timeit(lambda r: [(gemm(sm), passes(sh)) for _ in range(r)], [sm, sh], reps=3)
        g = timeit(lambda r: [gemm(sm) for _ in range(r)], [sm])
        h = timeit(lambda r: [passes(sh) for _ in range(r)], [sh])
        both = timeit(lambda r: [(gemm(sm), passes(sh)) for _ in range(r)], [sm, sh])
        print(f"  {256 - n:3d} + {n:3d} CUs: GEMM alone {1e3 * g:.0f} us ({g / g0:.2f}x), passes alone {1e3 * h:.0f} us ({h / h0:.2f}x), together {1e3 * both:.0f} us "
              f"(max of the two alone {1e3 * max(g, h):.0f}; unmasked serial {1e3 * (g0 + h0):.0f}, unmasked free streams {1e3 * both0:.0f})")


def launch_cost():
    """== 4. what a LAUNCH costs on a CU-masked stream: 2000 back-to-back tiny kernels (one workgroup, no work) per stream kind, and the same kernels
    alternating between two streams with an event edge per launch (the type-lane schedule of network.cu_partition)"""
    print("== 4. per-launch cost: 2000 empty one-workgroup kernels ==")
    out = torch.zeros(64, dtype=torch.int32, device=dev)
    n = 2000

    def chain(s):
        def run(reps):
            for _ in range(reps):
                for _ in range(n):
                    probe.cu_where(out.data_ptr(), 1, 0, s.cuda_stream)
        return run
    plain = torch.cuda.Stream()
    masked = cu_masked_stream(partition_masks(64)[0])
    full_mask = cu_masked_stream([0xFFFFFFFF] * 8)
    for label, s in (("unmasked stream", plain), ("CU-masked stream (192 CUs)", masked), ("CU-masked stream, all 256 bits set", full_mask)):
        chain(s)(1)
        ms = timeit(chain(s), [s], reps=3)
        print(f"  {label:40s} {1e3 * ms / n:6.2f} us per launch")

    def pingpong(sa, sb):
        evs = [torch.cuda.Event() for _ in range(n)]

        def run(reps):
            for _ in range(reps):
                for k in range(n):
                    s, o = (sa, sb) if k & 1 else (sb, sa)
                    if k:
                        s.wait_event(evs[k - 1])
                    probe.cu_where(out.data_ptr(), 1, 0, s.cuda_stream)
                    evs[k].record(s)
        return run
    for label, sa, sb in (("two unmasked streams, event per launch", plain, torch.cuda.Stream()),
                          ("two CU-masked streams (192 / 64), event per launch", masked, cu_masked_stream(partition_masks(64)[1]))):
        pingpong(sa, sb)(1)
        ms = timeit(pingpong(sa, sb), [sa, sb], reps=3)
        print(f"  {label:52s} {1e3 * ms / n:6.2f} us per launch")


if __name__ == "__main__":
    if LAUNCH_COST:
        launch_cost()
    else:
        main()
