"""Error of SPLIT-PRECISION matrix products (bf16 pieces of both fp32 operands on the bf16 MFMA, fp32 accumulation) inside the shipped convolution
forms -- the error study VERDICT r4 next-8 asks for BEFORE any split-precision kernel is written (a labelled variant, never the headline).

Same protocol as tools/wino2d_fm5_error.py: one Cout channel, weight transforms in fp64 rounded once to fp32 (packed offline), data transforms and the
output transform in fp32 in kernel order, reference = fp64 direct form.  Only the product accumulation over Cin changes:
    fp32      M += u * v                                      (the shipped kernels: v_mfma_f32_32x32x2_f32)
    bf16x1    M += bf(u) * bf(v)                              (plain bf16 operands: 1 bf16 MFMA per product, 16x the fp32 matrix rate)
    bf16x3    u = u0 + u1, v = v0 + v1 (two bf16 pieces each = 16 mantissa bits): u0 v0 + u0 v1 + u1 v0                 (3 MFMAs: 5.3x the rate)
    bf16x6    three pieces each (24 bits): u0 v0 + u0 v1 + u1 v0 + u1 v1 + u0 v2 + u2 v0                                  (6 MFMAs: 2.7x the rate)
A bf16 x bf16 product is exact in fp32; the terms are accumulated in fp32, smallest first, as separate MFMA passes would.
    python tools/bf16split_error.py [Cin ...]
"""
import sys

import numpy as np

from wino2d_error import f32, toom
from wino2d_fm5_error import PT4, PT8, apply_rows


def bf16(a):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def pieces(a, n):
    out, rest = [], np.asarray(a, np.float32)
    for _ in range(n):
        p = bf16(rest)
        out.append(p)
        rest = (rest - p).astype(np.float32)            # exact in fp32
    return out


TERMS = {"bf16x1": (1, [(0, 0)]), "bf16x3": (2, [(1, 0), (0, 1), (0, 0)]), "bf16x6": (3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])}


def mac(U, V, mode):
    """sum over the leading (Cin) axis of U[c] * V[c] (broadcast), fp32 accumulation channel by channel"""
    acc = np.zeros(np.broadcast_shapes(U.shape[1:], V.shape[1:]), np.float32)
    if mode == "fp32":
        for c in range(U.shape[0]):
            acc += U[c] * V[c]
        return acc
    n, terms = TERMS[mode]
    up, vp = pieces(U, n), pieces(V, n)
    for i, j in terms:                                    # one MFMA pass per term over the whole K axis, smallest terms first
        for c in range(U.shape[0]):
            acc += up[i][c] * vp[j][c]
    return acc


def run(Cin, form, mode, seed=0, nF=4, nT=6):
    rng = np.random.default_rng(seed)
    mF, mT = {"direct": (0, 0), "F(4,3)": (0, 4), "F(8,3)": (0, 8), "F(4,5)xF(4,3)": (4, 4)}[form]
    pT = PT8 if mT == 8 else PT4
    Ft, Tt = (mF or 4) * nF, (mT or 8) * nT
    x = rng.standard_normal((Cin, Ft + 4, Tt + 2))
    x = 0.5 * x * (1 + np.tanh(0.79788456 * (x + 0.044715 * x ** 3)))
    x = f32(x * (1 + 0.3 * rng.standard_normal((Cin, 1, 1))))
    w = f32(rng.standard_normal((Cin, 5, 3)) / np.sqrt(Cin * 15))
    xd, wd = x.astype(np.float64), w.astype(np.float64)
    ref = np.zeros((Ft, Tt))
    for kh in range(5):
        for kw in range(3):
            ref += np.einsum("c,cft->ft", wd[:, kh, kw], xd[:, kh:kh + Ft, kw:kw + Tt])
    if not mT:
        U = np.stack([w[:, kh, kw] for kh in range(5) for kw in range(3)], 1)[:, :, None, None]                      # [Cin, 15, 1, 1]
        V = np.stack([x[:, kh:kh + Ft, kw:kw + Tt] for kh in range(5) for kw in range(3)], 1)                          # [Cin, 15, F, T]
        y = mac(U, V, mode).sum(0, dtype=np.float32)
    else:
        ATt, Gt, BTt = toom(pT, mT, 3)
        n_t = mT + 2
        xt = np.lib.stride_tricks.sliding_window_view(x, n_t, axis=2)[:, :, ::mT][:, :, :nT]
        V = apply_rows(BTt, np.ascontiguousarray(xt), 3)                                                               # [Cin, F+4, nT, n_t]
        if not mF:
            U = f32(np.einsum("xk,chk->chx", Gt, wd))                                                                  # [Cin, 5, n_t]
            M = np.zeros((Ft, nT, n_t), np.float32)
            for kh in range(5):
                M += mac(U[:, kh][:, None, None, :], V[:, kh:kh + Ft], mode)
            y = apply_rows(ATt, M, 2).reshape(Ft, Tt)
        else:
            ATf, Gf, BTf = toom([0, 1, -1, 2, -2, 0.5, -0.5], mF, 5)
            n_f = mF + 4
            U2 = f32(np.einsum("yh,xk,chk->cyx", Gf, Gt, wd))
            Vr = np.lib.stride_tricks.sliding_window_view(V, n_f, axis=1)[:, ::mF][:, :nF]
            V2 = apply_rows(BTf, np.ascontiguousarray(Vr), 4)                                                          # [Cin, nF, nT, n_t, n_f]
            M2 = mac(np.ascontiguousarray(U2.transpose(0, 2, 1))[:, None, None], V2, mode)
            Y1 = apply_rows(ATf, M2, 3)
            Y2 = apply_rows(ATt, Y1, 2)
            y = Y2.transpose(0, 3, 1, 2).reshape(Ft, Tt)
    e = y.astype(np.float64) - ref
    return np.linalg.norm(e) / np.linalg.norm(ref)


def trunc_pieces(a):
    """the KERNEL's split (csrc/aid_wino2d.hip w2d_split3): truncation to the top 16 bits, three times; x = p0 + p1 + p2 exactly"""
    out, rest = [], np.asarray(a, np.float32)
    for _ in range(3):
        p = (rest.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        out.append(p)
        rest = (rest - p).astype(np.float32)
    assert not rest.any()
    return out


def per_product(n=100000, seed=0):
    """|dropped terms| / |product| of ONE fp32 x fp32 product under the six-product form (kept: p0q0 p0q1 p1q0 p1q1 p0q2 p2q0; dropped: p1q2, p2q1 -- each
    ~2^-24 of the product -- and p2q2 ~2^-32).  VERDICT r5 weak-3: the dropped part is NOT below 2^-32; it is of the order of one fp32 rounding."""
    rng = np.random.default_rng(seed)
    x, y = f32(rng.standard_normal(n)), f32(rng.standard_normal(n))
    p, q = [v.astype(np.float64) for v in trunc_pieces(x)], [v.astype(np.float64) for v in trunc_pieces(y)]
    prod = x.astype(np.float64) * y.astype(np.float64)
    rows = []
    for name, dropped in (("six products (the kernel)", p[1] * q[2] + p[2] * q[1] + p[2] * q[2]), ("eight products (not built)", p[2] * q[2])):
        r = np.abs(dropped) / np.abs(prod)
        rows.append((name, r.max(), np.median(r)))
        print(f"  {name:28s} dropped / product: max {r.max():.2e} = 2^{np.log2(r.max()):.1f}, median 2^{np.log2(np.median(r)):.1f}   (one fp32 rounding: 2^-24)")
    return rows


if __name__ == "__main__":
    if "--per-product" in sys.argv:
        print("per-product truncation of the split-precision variant (1e5 random fp32 pairs, the kernel's truncation split):")
        per_product()
        sys.exit(0)
    cins = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
    print("rel-L2 error of one conv layer (mean of 3 seeds); budget per layer: 1e-5.  MFMA cost per output relative to the shipped fp32 kernel of the same form:")
    print("bf16x1 1/16, bf16x3 3/16, bf16x6 6/16 (dense bf16 MFMA = 16x the fp32 matrix rate on gfx950)")
    for Cin in cins:
        print(f"Cin = {Cin}")
        for form in ("direct", "F(4,3)", "F(8,3)", "F(4,5)xF(4,3)"):
            row = []
            for mode in ("fp32", "bf16x6", "bf16x3", "bf16x1"):
                row.append(np.mean([run(Cin, form, mode, seed=s) for s in range(3)]))
            print(f"  {form:16s} fp32 {row[0]:.2e}   bf16x6 {row[1]:.2e}   bf16x3 {row[2]:.2e}   bf16x1 {row[3]:.2e}")
