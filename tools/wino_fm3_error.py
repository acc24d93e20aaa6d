"""fp32 error of larger 1-D Winograd forms F(m,3) along T for the dilated 5x3 convolution.

Same protocol as tools/wino2d_error.py: one Cout channel, K = Cin*15 reduction, weights' transforms in fp64 rounded
once (packed offline), data transform / accumulation over (ci, kh) / output transform in fp32 in kernel order;
reference = fp64 direct form.  Compares direct fp32, F(4,3), F(6,3) and F(8,3) for several interpolation point sets.
CPU only, a few seconds.
"""
import sys

import numpy as np

from wino2d_error import f32, toom


def run(Cin, m, points, seed=0, Tt=192, Ft=8, scale_rows=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((Cin, Ft + 4, Tt + 2))
    x = 0.5 * x * (1 + np.tanh(0.79788456 * (x + 0.044715 * x ** 3)))
    x = f32(x * (1 + 0.3 * rng.standard_normal((Cin, 1, 1))))
    w = f32(rng.standard_normal((Cin, 5, 3)) / np.sqrt(Cin * 15))
    xd, wd = x.astype(np.float64), w.astype(np.float64)
    ref = np.zeros((Ft, Tt))
    for kh in range(5):
        for kw in range(3):
            ref += np.einsum("c,cft->ft", wd[:, kh, kw], xd[:, kh:kh + Ft, kw:kw + Tt])
    if m == 0:
        acc = np.zeros((Ft, Tt), np.float32)
        for c in range(Cin):
            for kh in range(5):
                for kw in range(3):
                    acc += w[c, kh, kw] * x[c, kh:kh + Ft, kw:kw + Tt]
        y = acc
    else:
        n = m + 2
        AT, G, BT = toom(points, m, 3)
        if not scale_rows:                                   # undo toom()'s row balancing
            s = np.abs(BT).max(axis=1)
            AT, G, BT = toom(points, m, 3)
        U = f32(np.einsum("xk,chk->chx", G, wd))
        BT32, AT32 = f32(BT), f32(AT)
        ng = Tt // m
        xt = np.lib.stride_tricks.sliding_window_view(x, n, axis=2)[:, :, ::m][:, :, :ng]
        V = np.zeros(xt.shape[:3] + (n,), np.float32)
        for i in range(n):
            a = np.zeros(xt.shape[:3], np.float32)
            for k in range(n):
                if BT32[i, k] != 0:
                    a += BT32[i, k] * xt[..., k]
            V[..., i] = a
        M = np.zeros((Ft, ng, n), np.float32)
        for c in range(Cin):
            for kh in range(5):
                M += U[c, kh][None, None, :] * V[c, kh:kh + Ft]
        y = np.zeros((Ft, ng, m), np.float32)
        for o in range(m):
            for i in range(n):
                if AT32[o, i] != 0:
                    y[..., o] += AT32[o, i] * M[..., i]
        y = y.reshape(Ft, Tt)
    e = y.astype(np.float64) - ref
    return np.linalg.norm(e) / np.linalg.norm(ref), np.abs(e).max() / np.sqrt((ref ** 2).mean())


CASES = [
    ("direct fp32", 0, None),
    ("F(4,3) {0,+-1,+-2}", 4, [0, 1, -1, 2, -2]),
    ("F(4,3) {0,+-1,+-1/2}", 4, [0, 1, -1, 0.5, -0.5]),
    ("F(6,3) {0,+-1,+-2,+-1/2}", 6, [0, 1, -1, 2, -2, 0.5, -0.5]),
    ("F(6,3) {0,+-1,+-1/2,+-3/2}... ", 6, [0, 1, -1, 0.5, -0.5, 1.5, -1.5]),
    ("F(6,3) {0,+-1,+-2/3,+-3/2}", 6, [0, 1, -1, 2 / 3, -2 / 3, 1.5, -1.5]),
    ("F(8,3) {0,+-1,+-2,+-1/2,+-4}", 8, [0, 1, -1, 2, -2, 0.5, -0.5, 4, -4]),
    ("F(8,3) {0,+-1,+-2,+-1/2,+-1/4}", 8, [0, 1, -1, 2, -2, 0.5, -0.5, 0.25, -0.25]),
    ("F(8,3) {0,+-1,+-2,+-1/2,+-3/2}", 8, [0, 1, -1, 2, -2, 0.5, -0.5, 1.5, -1.5]),
    ("F(8,3) {0,+-1,+-1/2,+-2,+-3/4}... ", 8, [0, 1, -1, 0.5, -0.5, 2, -2, 0.75, -0.75]),
    ("F(8,3) {0,+-1/2,+-1,+-3/2,+-2/3}", 8, [0, 0.5, -0.5, 1, -1, 1.5, -1.5, 2 / 3, -2 / 3]),
    ("F(8,3) chebyshev-like", 8, [0, 0.4, -0.4, 0.8, -0.8, 1.25, -1.25, 2.5, -2.5]),
]

if __name__ == "__main__":
    cins = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
    for Cin in cins:
        print(f"Cin={Cin}")
        for name, m, pts in CASES:
            Tt = 192 if m in (0, 4, 6, 8) else 192
            r = [run(Cin, m, pts, seed=s, Tt=Tt) for s in range(3)]
            print(f"    {name:36s} rel-L2 {np.mean([x[0] for x in r]):.2e}   max/rms {np.max([x[1] for x in r]):.2e}")
