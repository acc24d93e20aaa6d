"""fp32 error budget of a 2-D Winograd form for the dilated 5x3 convolution.

Compares, on one output row-pair strip of a K = Cin*15 reduction with fp32 arithmetic emulated in numpy:
  direct fp32 | F(4,3) along T (what conv53_wino4* compute) | F(2,5) along F x F(4,3) along T (36 products per 2x4 outputs).
Weights' transforms are computed in fp64 and rounded once (they are packed offline); data transforms, the
products' accumulation over Cin and the output transforms run in fp32, in the order a kernel would use.
Reference: fp64 direct form.  Prints rel-L2 and max-abs/rms errors.  CPU only, a few seconds.
"""
import numpy as np


def toom(points, m, r):
    """Cook-Toom matrices (A^T [m,n], G [n,r], B^T [n,n]) for F(m,r), n = m+r-1, finite points + infinity."""
    n = m + r - 1
    p = np.asarray(points, dtype=np.float64)
    assert len(p) == n - 1

    def vander(cols):
        V = np.zeros((n, cols))
        for j in range(n - 1):
            V[j] = p[j] ** np.arange(cols)
        V[n - 1, cols - 1] = 1.0
        return V
    AT = vander(m).T
    G = vander(r)
    BT = np.linalg.inv(vander(n)).T
    # balance: move each point's scale so that B^T has small integers where possible (max |row| = 1 normalisation of G)
    s = np.abs(BT).max(axis=1)
    return AT, G * s[:, None], BT / s[:, None]


def f32(a):
    return np.asarray(a, dtype=np.float32)


def run(Cin, points_T, points_F, seed=0, act=True):
    rng = np.random.default_rng(seed)
    Tt, Ft = 64, 8                                            # outputs: 8 rows x 64 columns, one Cout channel
    x = rng.standard_normal((Cin, Ft + 4, Tt + 2))
    if act:
        x = 0.5 * x * (1 + np.tanh(0.79788456 * (x + 0.044715 * x ** 3)))
    x = f32(x * (1 + 0.3 * rng.standard_normal((Cin, 1, 1))))   # modulated, fp32 activations
    w = f32(rng.standard_normal((Cin, 5, 3)) / np.sqrt(Cin * 15))
    xd, wd = x.astype(np.float64), w.astype(np.float64)
    ref = np.zeros((Ft, Tt))
    for kh in range(5):
        for kw in range(3):
            ref += np.einsum("c,cft->ft", wd[:, kh, kw], xd[:, kh:kh + Ft, kw:kw + Tt])
    out = {}
    # direct fp32 (sequential accumulation over ci, kh, kw)
    acc = np.zeros((Ft, Tt), np.float32)
    for c in range(Cin):
        for kh in range(5):
            for kw in range(3):
                acc += w[c, kh, kw] * x[c, kh:kh + Ft, kw:kw + Tt]
    out["direct fp32"] = acc
    # F(4,3) along T
    AT, G, BT = toom(points_T, 4, 3)
    U = f32(np.einsum("xk,chk->chx", G, wd))                  # [Cin,5,6]
    BT32, AT32 = f32(BT), f32(AT)
    xt = np.lib.stride_tricks.sliding_window_view(x, 6, axis=2)[:, :, ::4][:, :, :Tt // 4]   # [Cin,F+4,G,6]
    V = np.zeros(xt.shape[:3] + (6,), np.float32)
    for i in range(6):
        a = np.zeros(xt.shape[:3], np.float32)
        for k in range(6):
            if BT32[i, k] != 0:
                a += BT32[i, k] * xt[..., k]
        V[..., i] = a
    M = np.zeros((Ft, Tt // 4, 6), np.float32)
    for c in range(Cin):
        for kh in range(5):
            M += U[c, kh][None, None, :] * V[c, kh:kh + Ft]
    y = np.zeros((Ft, Tt // 4, 4), np.float32)
    for o in range(4):
        for i in range(6):
            if AT32[o, i] != 0:
                y[..., o] += AT32[o, i] * M[..., i]
    out["F(4,3) along T"] = y.reshape(Ft, Tt)
    # F(2,5) along F x F(4,3) along T
    ATf, Gf, BTf = toom(points_F, 2, 5)
    U2 = f32(np.einsum("yh,xk,chk->cyx", Gf, G, wd))          # [Cin,6,6]
    BTf32, ATf32 = f32(BTf), f32(ATf)
    # F transform of V (already T-transformed): rows 2p .. 2p+5
    P = Ft // 2
    V2 = np.zeros((Cin, P, 6, Tt // 4, 6), np.float32)
    for pidx in range(P):
        for i in range(6):
            a = np.zeros((Cin, Tt // 4, 6), np.float32)
            for k in range(6):
                if BTf32[i, k] != 0:
                    a += BTf32[i, k] * V[:, 2 * pidx + k]
            V2[:, pidx, i] = a
    M2 = np.zeros((P, 6, Tt // 4, 6), np.float32)
    for c in range(Cin):
        M2 += U2[c][None, :, None, :] * V2[c]
    y2 = np.zeros((P, 2, Tt // 4, 4), np.float32)
    for of in range(2):
        tmp = np.zeros((P, Tt // 4, 6), np.float32)
        for i in range(6):
            if ATf32[of, i] != 0:
                tmp += ATf32[of, i] * M2[:, i]
        for o in range(4):
            for i in range(6):
                if AT32[o, i] != 0:
                    y2[:, of, :, o] += AT32[o, i] * tmp[..., i]
    out["F(2,5) x F(4,3)"] = y2.transpose(0, 1, 2, 3).reshape(Ft, Tt // 4, 4).reshape(Ft, Tt)
    res = {}
    for k, v in out.items():
        e = v.astype(np.float64) - ref
        res[k] = (np.linalg.norm(e) / np.linalg.norm(ref), np.abs(e).max() / np.sqrt((ref ** 2).mean()))
    return res


if __name__ == "__main__":
    PT = [0, 1, -1, 2, -2]
    for Cin in (64, 96, 128, 256):
        for name, PF in (("{0,+-1,+-2,inf}", [0, 1, -1, 2, -2]), ("{0,+-1,+-1/2,inf}", [0, 1, -1, 0.5, -0.5])):
            r = [run(Cin, PT, PF, seed=s) for s in range(3)]
            print(f"Cin={Cin:4d}  F-points {name}")
            for k in r[0]:
                print(f"    {k:18s} rel-L2 {np.mean([x[k][0] for x in r]):.2e}   max/rms {np.max([x[k][1] for x in r]):.2e}")
