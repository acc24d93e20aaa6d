"""fp32 error of NON-FUSED 2-D Winograd forms F(mF,5) along the dilated axis x F(mT,3) along T for the 5x3 convolution
("next (0)": the batched-GEMM form for the C >= 128 layers; VERDICT r3 next-1 follow-up).

Same protocol as tools/wino2d_error.py / tools/wino_fm3_error.py: one Cout channel, weights' transforms in fp64 rounded once
(packed offline), data transform (T axis first, then the row axis), the products' accumulation over Cin and the output
transform (row axis first, then T) in fp32 in the order a kernel would use; reference = fp64 direct form.
Products per output: F(mF,5) x F(mT,3) = (mF+4)(mT+2) / (mF mT)  against 15 for the direct form and 3.75 for F(8,3) alone.
The dilation does not enter (a tile walks one residue class of rows).  CPU only, about a minute.
    python tools/wino2d_fm5_error.py [Cin ...]         # --search: point-set search for the row axis
"""
import itertools
import sys

import numpy as np

from wino2d_error import f32, toom

PT8 = [0, 0.4, -0.4, 0.8, -0.8, 1.25, -1.25, 2.5, -2.5]          # the shipped F(8,3) points (csrc/aid_wino8.h)
PT4 = [0, 1, -1, 2, -2]


def apply_rows(Mx, X, axis):
    """fp32 y = Mx . X along `axis`, accumulated term by term (skipping exact zeros) as a kernel's FMA chain does."""
    Mx = f32(Mx)
    X = np.moveaxis(X, axis, 0)
    out = np.zeros((Mx.shape[0],) + X.shape[1:], np.float32)
    for i in range(Mx.shape[0]):
        a = np.zeros(X.shape[1:], np.float32)
        for k in range(Mx.shape[1]):
            if Mx[i, k] != 0:
                a += Mx[i, k] * X[k]
        out[i] = a
    return np.moveaxis(out, 0, axis)


def run(Cin, mF, pF, mT, pT, seed=0, nF=4, nT=6):
    """nF x nT tiles of mF x mT outputs.  mF = 0: 1-D form along T only (the shipped kernels); mT = 0 and mF = 0: direct fp32."""
    rng = np.random.default_rng(seed)
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
        acc = np.zeros((Ft, Tt), np.float32)
        for c in range(Cin):
            for kh in range(5):
                for kw in range(3):
                    acc += w[c, kh, kw] * x[c, kh:kh + Ft, kw:kw + Tt]
        y = acc
    else:
        ATt, Gt, BTt = toom(pT, mT, 3)
        n_t = mT + 2
        xt = np.lib.stride_tricks.sliding_window_view(x, n_t, axis=2)[:, :, ::mT][:, :, :nT]          # [Cin, F+4, nT, n_t]
        V = apply_rows(BTt, np.ascontiguousarray(xt), 3)                                               # T-transformed rows
        if not mF:
            U = f32(np.einsum("xk,chk->chx", Gt, wd))
            M = np.zeros((Ft, nT, n_t), np.float32)
            for c in range(Cin):
                for kh in range(5):
                    M += U[c, kh][None, None, :] * V[c, kh:kh + Ft]
            y = apply_rows(ATt, M, 2).reshape(Ft, Tt)
        else:
            ATf, Gf, BTf = toom(pF, mF, 5)
            n_f = mF + 4
            U2 = f32(np.einsum("yh,xk,chk->cyx", Gf, Gt, wd))                                           # [Cin, n_f, n_t]
            Vr = np.lib.stride_tricks.sliding_window_view(V, n_f, axis=1)[:, ::mF][:, :nF]              # [Cin, nF, nT, n_t, n_f]
            V2 = apply_rows(BTf, np.ascontiguousarray(Vr), 4)                                           # [Cin, nF, nT, n_t, n_f]
            M2 = np.zeros(V2.shape[1:], np.float32)
            U2t = np.ascontiguousarray(U2.transpose(0, 2, 1))                                           # [Cin, n_t, n_f]
            for c in range(Cin):
                M2 += U2t[c][None, None] * V2[c]
            Y1 = apply_rows(ATf, M2, 3)                                                                 # [nF, nT, n_t, mF]
            Y2 = apply_rows(ATt, Y1, 2)                                                                 # [nF, nT, mT, mF]
            y = Y2.transpose(0, 3, 1, 2).reshape(Ft, Tt)
    e = y.astype(np.float64) - ref
    return np.linalg.norm(e) / np.linalg.norm(ref), np.abs(e).max() / np.sqrt((ref ** 2).mean())


def products(mF, mT):
    return ((mF + 4) if mF else 5) * (mT + 2) / ((mF or 1) * mT) if mT else 15.0


CASES = [
    ("direct fp32", 0, None, 0, None),
    ("F(4,3) along T (shipped)", 0, None, 4, PT4),
    ("F(8,3) along T (shipped)", 0, None, 8, PT8),
    ("F(2,5){0,+-1,+-1/2} x F(4,3)   [r02 study]", 2, [0, 1, -1, 0.5, -0.5], 4, PT4),
    ("F(2,5){0,+-1,+-1/2} x F(8,3)", 2, [0, 1, -1, 0.5, -0.5], 8, PT8),
    ("F(2,5){0,+-0.6,+-1.4} x F(8,3)", 2, [0, 0.6, -0.6, 1.4, -1.4], 8, PT8),
    ("F(3,5){0,+-1,+-1/2,2} x F(8,3)", 3, [0, 1, -1, 0.5, -0.5, 2], 8, PT8),
    ("F(4,5){0,+-1,+-2,+-1/2} x F(8,3)", 4, [0, 1, -1, 2, -2, 0.5, -0.5], 8, PT8),
    ("F(4,5){0,+-0.5,+-1,+-2} scaled 0.8 x F(8,3)", 4, [0, 0.4, -0.4, 0.8, -0.8, 1.6, -1.6], 8, PT8),
    ("F(4,5){0,+-0.6,+-1,+-1/0.6} x F(8,3)", 4, [0, 0.6, -0.6, 1, -1, 1 / 0.6, -1 / 0.6], 8, PT8),
    ("F(4,5){0,+-1,+-2,+-1/2} x F(4,3)", 4, [0, 1, -1, 2, -2, 0.5, -0.5], 4, PT4),
]


def search(Cin=128):
    """Symmetric point sets {0, +-a, +-b, +-c} for F(4,5) (and {0, +-a, +-b} for F(2,5)) on the row axis, F(8,3) on T."""
    grid = [0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 1.0, 1.25, 1.5, 5 / 3, 2.0, 2.5, 3.0]
    best = []
    for a, b, c in itertools.combinations(grid, 3):
        pts = [0, a, -a, b, -b, c, -c]
        r = np.mean([run(Cin, 4, pts, 8, PT8, seed=s, nF=2, nT=3)[0] for s in range(2)])
        best.append((r, (a, b, c)))
    best.sort()
    print(f"F(4,5) x F(8,3), Cin={Cin}: best symmetric row-axis point sets of {len(best)}")
    for r, p in best[:8]:
        print(f"    +-{p}: rel-L2 {r:.2e}")
    print(f"    worst: +-{best[-1][1]}: {best[-1][0]:.2e}")
    best2 = []
    for a, b in itertools.combinations(grid, 2):
        pts = [0, a, -a, b, -b]
        r = np.mean([run(Cin, 2, pts, 8, PT8, seed=s, nF=4, nT=3)[0] for s in range(2)])
        best2.append((r, (a, b)))
    best2.sort()
    print(f"F(2,5) x F(8,3), Cin={Cin}: best of {len(best2)}")
    for r, p in best2[:5]:
        print(f"    +-{p}: rel-L2 {r:.2e}")
    return best[0][1], best2[0][1]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cins = [int(a) for a in args] or [128, 256]
    cases = list(CASES)
    if "--search" in sys.argv:
        p4, p2 = search()
        cases.append((f"F(4,5) searched +-{p4} x F(8,3)", 4, [0] + [s * v for v in p4 for s in (1, -1)], 8, PT8))
        cases.append((f"F(2,5) searched +-{p2} x F(8,3)", 2, [0] + [s * v for v in p2 for s in (1, -1)], 8, PT8))
    for Cin in cins:
        print(f"Cin={Cin}   (budget per layer: 1e-5)")
        for name, mF, pF, mT, pT in cases:
            r = [run(Cin, mF, pF, mT, pT, seed=s) for s in range(3)]
            print(f"    {name:48s} {products(mF, mT):5.2f} products/output   rel-L2 {np.mean([x[0] for x in r]):.2e}   max/rms {np.max([x[1] for x in r]):.2e}")
