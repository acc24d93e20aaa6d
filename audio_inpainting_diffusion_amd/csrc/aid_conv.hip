// aid_conv2d: fused dilated dense convolution as an fp32-MFMA implicit GEMM for gfx950.
//
//   GEMM view:  M = Cout, N = (b, f, t) output positions, K = (ci, kh, kw).
//   One workgroup (4 waves, one per SIMD) computes an M_BLK x N_BLK tile; its N tile is ROWS consecutive (b,f)
//   rows times TT consecutive t (TT = min(N_BLK, pow2ceil(T)), ROWS = N_BLK/TT), so the three kw taps of a row
//   come from one LDS strip with a 1-sample halo each side and the five kh taps are five dilated input rows.
//   K is walked in chunks of KC input channels through a DOUBLE-BUFFERED LDS tile:
//     x strips  [KC][ROWS][KH][TT+8]   global -> registers -> (scale, GELU) -> LDS
//     weights   [KH*KW][KC][M_BLK]     pre-packed (cout contiguous), float4 copies
//   Software pipeline (round 1, v2): the global loads of chunk c+1 are issued into registers BEFORE the MFMA
//   loop of chunk c; their transform + ds_write into the other LDS buffer is interleaved slot by slot with
//   the MFMAs (VALU / LDS-write issue next to the matrix pipe), and ONE barrier closes the chunk.  All
//   per-thread staging addresses are computed once, outside the K loop.  (v1 staged synchronously with
//   runtime-trip-count loops: one load in flight per wave, ~20 k cycles of exposed latency per chunk = 41 %
//   of the fp32 MFMA peak; see profiles/r01_*.)
//   MFMA: v_mfma_f32_32x32x2_f32, A = W[32 cout][2 ci], B = X[2 ci][32 t]; both operands are conflict-free
//   ds_read_b32 (lanes 0-31 consecutive floats; the two half-waves read different ci).
//   Epilogue (registers -> global): y = alpha * (res_scale*res + acc*out_scale[b,co])  or the dGELU form (epi=1).
//
//   fp32 MFMA is exact fp32 FMA arithmetic (cdna_hip_programming.md section 3), so parity with the reference's
//   F.conv2d is limited only by summation order.
#include "aid_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDev {
    aid_conv2d_params p;
    int tt_log2;      // log2(TT)
    int rows_log2;    // log2(ROWS)
    int tiles_t;      // ceil(T / TT)
    int nrows;        // B*F
    int nchunks;      // ceil(Cin / KC)
    int splits, cps;  // split-K: blockIdx.z owns chunks [z*cps, min(nchunks, (z+1)*cps)); splits = 1: whole K, normal epilogue
    float* ws;        // [splits][B][Cout][F][T] partial sums when splits > 1
};

__device__ __forceinline__ float4 xform4(float4 v, float sc, int act) {
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    if (act == 1) { v.x = aid_gelu(v.x); v.y = aid_gelu(v.y); v.z = aid_gelu(v.z); v.w = aid_gelu(v.w); }
    return v;
}

template <int KH, int KW, int MT, int NT, int WGM, int WGN, int KC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_mfma_kernel(const ConvDev a) {
    constexpr int NTHREADS = 64 * WGM * WGN;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    constexpr int TAPS = KH * KW;
    constexpr int PADL = (KW > 1) ? 4 : 0;                   // strip: [3]=left halo, [4..4+TT) core, [4+TT]=right halo
    constexpr int XOFF = PADL - KW / 2;
    constexpr int XQ_TOTAL = KC * KH * (N_BLK / 4);
    constexpr int XQ = (XQ_TOTAL + NTHREADS - 1) / NTHREADS; // x float4s per thread per chunk
    constexpr int MQ = M_BLK / 4;
    constexpr int WQ_TOTAL = TAPS * KC * MQ;
    constexpr int WQ = (WQ_TOTAL + NTHREADS - 1) / NTHREADS; // weight float4s per thread per chunk
    constexpr int HQ = 2;                                    // halo elements per thread per chunk (fast path)
    constexpr int KSTEPS = TAPS * (KC / 2);
    constexpr int NSLOT = (KH > 1) ? KH : (KSTEPS >= 4 ? 4 : KSTEPS);
    constexpr int KS_PER_SLOT = KSTEPS / NSLOT;
    static_assert(KSTEPS % NSLOT == 0 && NSLOT >= 2, "k-steps must split evenly into >= 2 slots");

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;
    const int TTP = TT + 2 * PADL;
    const int XSZ = KC * ROWS * KH * TTP;
    constexpr int WSZ = TAPS * KC * M_BLK;
    const int BUFSZ = XSZ + WSZ;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* rowinfo = (int*)(smem + 2 * BUFSZ);                 // [ROWS][2] = (b, f) or b = -1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN;
    const int wn = wave % WGN;

    const int tile_t = blockIdx.x % a.tiles_t;
    const int rg = blockIdx.x / a.tiles_t;
    const int row0 = rg << a.rows_log2;
    const int t0 = tile_t << a.tt_log2;
    const int m0 = blockIdx.y * M_BLK;

    for (int r = tid; r < ROWS; r += NTHREADS) {
        const int rid = row0 + r;
        int b = -1, f = 0;
        if (rid < a.nrows) { b = rid / p.F; f = rid - b * p.F; }
        rowinfo[2 * r] = b;
        rowinfo[2 * r + 1] = f;
    }
    __syncthreads();

    // ---- per-thread staging descriptors (constant over the K loop) ------------------------------------------
    const int tq_log2 = a.tt_log2 - 2;
    int64_t xg[XQ];      // global element offset without the channel term, -1 = out of range
    int xlds[XQ];        // LDS float offset inside the X tile
    int xci[XQ];         // channel within the chunk
    int xsc[XQ];         // in_scale row offset b * in_scale_ld
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
        const int q = tid + i * NTHREADS;
        const int tq = q & ((1 << tq_log2) - 1);
        const int s = q >> tq_log2;
        const int kh = s % KH;
        const int s2 = s / KH;
        const int rr = s2 & (ROWS - 1);
        const int ci = s2 >> a.rows_log2;
        const int b = rowinfo[2 * rr];
        const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
        const int t = t0 + 4 * tq;
        const bool ok = b >= 0 && fi >= 0 && fi < p.F && t < p.T && q < XQ_TOTAL;
        xg[i] = ok ? ((int64_t)b * p.x.sB + (int64_t)fi * p.x.sF + t) : -1;
        xlds[i] = (q < XQ_TOTAL) ? (s * TTP + PADL + 4 * tq) : -1;
        xci[i] = ci;
        xsc[i] = ok ? (int)(b * p.in_scale_ld) : 0;
    }
    const int nstrips = KC * ROWS * KH;
    const bool halo_fast = (KW > 1) && (2 * nstrips <= HQ * NTHREADS);
    int64_t hg[HQ];
    int hlds[HQ], hci[HQ], hsc[HQ];
#pragma unroll
    for (int i = 0; i < HQ; ++i) {
        hg[i] = -1; hlds[i] = -1; hci[i] = 0; hsc[i] = 0;
        const int h = tid + i * NTHREADS;
        if (KW > 1 && halo_fast && h < 2 * nstrips) {
            const int s = h >> 1, side = h & 1;
            const int kh = s % KH;
            const int s2 = s / KH;
            const int rr = s2 & (ROWS - 1);
            const int ci = s2 >> a.rows_log2;
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            const int t = side ? (t0 + TT) : (t0 - 1);
            const bool ok = b >= 0 && fi >= 0 && fi < p.F && t >= 0 && t < p.T;
            hg[i] = ok ? ((int64_t)b * p.x.sB + (int64_t)fi * p.x.sF + t) : -1;
            hlds[i] = s * TTP + (side ? 4 + TT : 3);
            hci[i] = ci;
            hsc[i] = ok ? (int)(b * p.in_scale_ld) : 0;
        }
    }
    int wg[WQ], wlds[WQ];    // weight source offset (without chunk / m0 terms) and LDS offset; -1 = idle slot
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        const int q = tid + i * NTHREADS;
        if (q < WQ_TOTAL) {
            const int m4 = q % MQ;
            const int ci = (q / MQ) % KC;
            const int tap = q / (MQ * KC);
            wg[i] = (tap * p.Cin_pad + ci) * p.Cout_pad + 4 * m4;
            wlds[i] = (tap * KC + ci) * M_BLK + 4 * m4;
        } else { wg[i] = -1; wlds[i] = -1; }
    }
    const float* wbase = p.wp + m0;

    // per-n-tile LDS offsets of this lane's B element (without the ci / kh / kw terms)
    int xoff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + (lane & 31);
        const int rr = n >> a.tt_log2;
        const int tt = n & (TT - 1);
        xoff[j] = rr * KH * TTP + tt + XOFF;
    }
    const int ci_lane = lane >> 5;
    const int strip_ci = ROWS * KH * TTP;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 xv[XQ];
    float xs[XQ];
    float hv[HQ];
    float4 wv[WQ];

    auto issue_loads = [&](int c0) {
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int c = c0 + xci[i];
            const bool ok = xg[i] >= 0 && c < p.Cin;
            xv[i] = ok ? *reinterpret_cast<const float4*>(p.x.p + xg[i] + (int64_t)c * p.x.sC) : make_float4(0.f, 0.f, 0.f, 0.f);
            xs[i] = (ok && p.in_scale) ? p.in_scale[xsc[i] + c] : 1.f;
        }
        if (KW > 1) {
#pragma unroll
            for (int i = 0; i < HQ; ++i) {
                const int c = c0 + hci[i];
                const bool ok = hg[i] >= 0 && c < p.Cin;
                float v = ok ? p.x.p[hg[i] + (int64_t)c * p.x.sC] : 0.f;
                if (ok && p.in_scale) v *= p.in_scale[hsc[i] + c];
                hv[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WQ; ++i)
            wv[i] = (wg[i] >= 0) ? *reinterpret_cast<const float4*>(wbase + wg[i] + (int64_t)c0 * p.Cout_pad)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // write the register-staged chunk into LDS buffer `Xs/Ws`; `slot` < 0 writes everything
    auto write_slot = [&](float* Xs, float* Ws, int slot) {
#pragma unroll
        for (int i = 0; i < XQ; ++i)
            if ((slot < 0 || (i % (NSLOT - 1)) == slot) && (XQ_TOTAL % NTHREADS == 0 || xlds[i] >= 0))
                *reinterpret_cast<float4*>(Xs + xlds[i]) = xform4(xv[i], xs[i], p.act);
#pragma unroll
        for (int i = 0; i < WQ; ++i)
            if ((slot < 0 || (i % (NSLOT - 1)) == slot) && wlds[i] >= 0)
                *reinterpret_cast<float4*>(Ws + wlds[i]) = wv[i];
        if (KW > 1 && (slot < 0 || slot == NSLOT - 2)) {
#pragma unroll
            for (int i = 0; i < HQ; ++i)
                if (hlds[i] >= 0) Xs[hlds[i]] = (p.act == 1) ? aid_gelu(hv[i]) : hv[i];
        }
    };
    // slow path for the halos of very short rows (only tiny test shapes): synchronous
    auto halo_slow = [&](float* Xs, int c0) {
        for (int h = tid; h < 2 * nstrips; h += NTHREADS) {
            const int s = h >> 1, side = h & 1;
            const int kh = s % KH;
            const int s2 = s / KH;
            const int rr = s2 & (ROWS - 1);
            const int ci = s2 >> a.rows_log2;
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            const int t = side ? (t0 + TT) : (t0 - 1);
            const int c = c0 + ci;
            float v = 0.f;
            if (b >= 0 && fi >= 0 && fi < p.F && c < p.Cin && t >= 0 && t < p.T) {
                v = p.x.p[(int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)fi * p.x.sF + t];
                if (p.in_scale) v *= p.in_scale[(int64_t)b * p.in_scale_ld + c];
                if (p.act == 1) v = aid_gelu(v);
            }
            Xs[s * TTP + (side ? 4 + TT : 3)] = v;
        }
    };

    // ---- prologue: first chunk of this workgroup's K range -> buffer 0 ------------------------------------------------
    const int cb = (int)blockIdx.z * a.cps;
    const int nloc = (a.nchunks - cb < a.cps) ? (a.nchunks - cb) : a.cps;
    issue_loads(cb * KC);
    write_slot(smem, smem + XSZ, -1);
    if (KW > 1 && !halo_fast) halo_slow(smem, cb * KC);
    __syncthreads();

    for (int ch = 0; ch < nloc; ++ch) {
        const int cur = ch & 1;
        const float* Xs = smem + cur * BUFSZ;
        const float* Ws = Xs + XSZ;
        float* Xn = smem + (cur ^ 1) * BUFSZ;
        float* Wn = Xn + XSZ;
        const bool more = (ch + 1) < nloc;
        // operand fragments are fetched one k-step AHEAD of the MFMAs that consume them (explicit register
        // double buffer), so every ds_read has a full k-step of matrix work to land behind
        float av[2][MT], bv[2][NT];
        auto load_frags = [&](int ks, int buf) {
            const int tap = ks / (KC / 2);
            const int cp = ks % (KC / 2);
            const int kh = tap / KW, kw = tap % KW;
            const int ci = 2 * cp + ci_lane;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                av[buf][i] = Ws[(tap * KC + ci) * M_BLK + (wm * MT + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bv[buf][j] = Xs[ci * strip_ci + kh * TTP + xoff[j] + kw];
        };
        load_frags(0, 0);
#pragma unroll
        for (int slot = 0; slot < NSLOT; ++slot) {
            // progress balancing: the hardware arbitrates the shared matrix pipe by priority, then AGE, so the
            // oldest wave of a SIMD finishes its chunk thousands of cycles before the youngest and then idles
            // at the barrier while the pipe runs under-occupied (measured with tools/conv_phase_probe.py).  A wave
            // lowers its own priority as it advances through the chunk, which lets the laggards catch up.
            if (slot == 0) __builtin_amdgcn_s_setprio(3);
            else if (slot == 1) __builtin_amdgcn_s_setprio(2);
            else if (slot == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int k = 0; k < KS_PER_SLOT; ++k) {
                const int ks = slot * KS_PER_SLOT + k;          // k-step index: (tap, cp) with cp fastest
                if (ks + 1 < KSTEPS) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
                // the next chunk's global loads are issued BEHIND the first MFMAs of this chunk, so the matrix
                // pipe is already busy while the wave spends its issue slots on address arithmetic / VMEM
                if (slot == 0 && k == 0 && more) issue_loads((cb + ch + 1) * KC);
            }
            if (more && slot > 0) write_slot(Xn, Wn, slot - 1);
        }
        if (more && KW > 1 && !halo_fast) halo_slow(Xn, (cb + ch + 1) * KC);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + (lane & 31);
        const int rr = n >> a.tt_log2;
        const int tt = n & (TT - 1);
        const int b = rowinfo[2 * rr];
        const int f = rowinfo[2 * rr + 1];
        const int t = t0 + tt;
        if (b < 0 || t >= p.T) continue;
        if (a.splits > 1) {                              // split-K: raw partial sums, reduced by conv_splitk_reduce_kernel
            const int64_t ft = (int64_t)p.F * p.T;
            float* wsb = a.ws + (((int64_t)blockIdx.z * p.B + b) * p.Cout) * ft + (int64_t)f * p.T + t;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mbase = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < p.Cout) wsb[(int64_t)m * ft] = acc[i][j][r];
                }
            }
            continue;
        }
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // gather phase first (res / aux / gate), then compute + store: res may alias y (gradient accumulation),
            // so the compiler would otherwise serialise load -> store element by element (16 dependent round trips)
            float rv[16], uv[16], sv[16];
            const int mbase = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                const bool ok = m < p.Cout;
                rv[r] = (ok && p.res.p) ? p.res.p[rbase + (int64_t)m * p.res.sC] : 0.f;
                sv[r] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                uv[r] = (ok && p.epi == 1) ? p.aux.p[abase + (int64_t)m * p.aux.sC] * p.aux_scale[(int64_t)b * p.aux_scale_ld + m] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m >= p.Cout) continue;
                float v = acc[i][j][r] * sv[r];
                if (p.epi == 1) v *= aid_dgelu(uv[r]);
                v += p.res_scale * rv[r];
                p.y.p[ybase + (int64_t)m * p.y.sC] = p.alpha * v;
            }
        }
    }
}

// y = alpha * (res_scale*res + out_scale * sum_z ws[z])  -- fixed summation order: deterministic
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const aid_conv2d_params p, const float* __restrict__ ws, int splits) {
    const int64_t ft = (int64_t)p.F * p.T;
    const int64_t q4 = ft >> 2;                           // float4s per (b, m) plane (T % 4 == 0)
    const int64_t total = (int64_t)p.B * p.Cout * q4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t bm = i / q4, e = (i - bm * q4) * 4;
    const int b = (int)(bm / p.Cout), m = (int)(bm - (int64_t)b * p.Cout);
    const int f = (int)(e / p.T), t = (int)(e - (int64_t)f * p.T);
    const int64_t plane = (int64_t)p.B * p.Cout * ft;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(ws + z * plane + bm * ft + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float sc = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res.p) r = *reinterpret_cast<const float4*>(p.res.p + (int64_t)b * p.res.sB + (int64_t)m * p.res.sC + (int64_t)f * p.res.sF + t);
    float4 o;
    o.x = p.alpha * (p.res_scale * r.x + s.x * sc); o.y = p.alpha * (p.res_scale * r.y + s.y * sc);
    o.z = p.alpha * (p.res_scale * r.z + s.z * sc); o.w = p.alpha * (p.res_scale * r.w + s.w * sc);
    *reinterpret_cast<float4*>(p.y.p + (int64_t)b * p.y.sB + (int64_t)m * p.y.sC + (int64_t)f * p.y.sF + t) = o;
}

int aid_conv_splitk_reduce(const aid_conv2d_params* p, const float* ws, int splits, hipStream_t st) {
    const int64_t total = (int64_t)p->B * p->Cout * ((int64_t)p->F * p->T / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *p, ws, splits);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
static int pick_mblk(int Cout) {
    if (Cout <= 32) return 32;
    if (Cout <= 64) return 64;
    if (Cout <= 96 || (Cout % 96 == 0 && Cout % 128 != 0)) return 96;   // 96, 192, 288: no padded rows
    return 128;
}

extern "C" void aid_conv2d_pack_dims(int Cin, int Cout, int* Cin_pad, int* Cout_pad) {
    const int mb = pick_mblk(Cout);
    *Cin_pad = ((Cin + 31) / 32) * 32;
    *Cout_pad = ((Cout + mb - 1) / mb) * mb;
}

template <int KH, int KW, int MT, int NT, int WGM, int WGN, int KC>
static int launch_cfg(const aid_conv2d_params* p, hipStream_t st, int splits = 1) {
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    ConvDev a;
    a.p = *p;
    int TT = aid_pow2ceil(p->T);
    if (TT > N_BLK) TT = N_BLK;
    if (TT < 4) TT = 4;
    a.tt_log2 = aid_ilog2(TT);
    const int ROWS = N_BLK / TT;
    a.rows_log2 = aid_ilog2(ROWS);
    a.tiles_t = aid_cdiv(p->T, TT);
    a.nrows = p->B * p->F;
    a.nchunks = aid_cdiv(p->Cin, KC);
    a.cps = aid_cdiv(a.nchunks, splits);
    a.splits = aid_cdiv(a.nchunks, a.cps);               // (no empty K ranges)
    a.ws = p->ws;
    const int rgroups = aid_cdiv(a.nrows, ROWS);
    dim3 grid((unsigned)(rgroups * a.tiles_t), (unsigned)(p->Cout_pad / M_BLK), (unsigned)a.splits);
    const size_t lds = sizeof(float) * 2 * ((size_t)KC * ROWS * KH * (TT + (KW > 1 ? 8 : 0)) + (size_t)KH * KW * KC * M_BLK) +
                       sizeof(int) * 2 * ROWS;
    AID_REQUIRE(lds <= 160 * 1024, "aid_conv2d: LDS tile too large");
    auto kern = conv_mfma_kernel<KH, KW, MT, NT, WGM, WGN, KC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel(a.splits > 1 ? "conv_mfma_kernel+splitk" : (KH == 5 ? "conv_mfma_kernel(5x3)" : "conv_mfma_kernel(1x1)"));
    if (a.splits > 1) {
        const int64_t total = (int64_t)p->B * p->Cout * ((int64_t)p->F * p->T / 4);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *p, (const float*)p->ws, a.splits);
        AID_CHECK_LAUNCH();
    }
    return AID_OK;
}

template <int KH, int KW, int KC>
static int launch_m(const aid_conv2d_params* p, hipStream_t st) {
    int mb = pick_mblk(p->Cout);
    if (mb == 128) {
        // grid-starved GEMMs (qk projections: N = B*T columns only)
        const int64_t npos = (int64_t)p->B * p->F * p->T;
        const int64_t ntiles = (npos + 255) / 256;
        const int64_t wg64 = ntiles * (p->Cout_pad / 64);
        if (KH == 1 && p->ws && p->epi == 0 && wg64 < 384 && p->Cin >= 16 * KC && p->Cout_pad % 64 == 0 && (p->T % 4) == 0) {
            // split K over several workgroups per 64 x 256 tile (deterministic two-pass reduction through ws)
            int S = (int)((768 + wg64 - 1) / wg64);
            if (S > 8) S = 8;
            const int64_t need = (int64_t)S * p->B * p->Cout * p->F * p->T * 4;
            auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
            if (S > 1 && need <= p->ws_bytes && al4(p->y) && (!p->res.p || al4(p->res)))
                return launch_cfg<KH, KW, 1, 2, 2, 4, KC>(p, st, S);
        }
        // single pass: trade tile height for more workgroups
        if (ntiles * (p->Cout_pad / 128) < 192) mb = (ntiles * (p->Cout_pad / 64) < 192) ? 32 : 64;
    }
    if (KH == 1) {                      // more waves per CU for the short-K (memory-bound) channel projections
        switch (mb) {
            case 32: return launch_cfg<KH, KW, 1, 2, 1, 4, KC>(p, st);
            case 64: return launch_cfg<KH, KW, 1, 2, 2, 4, KC>(p, st);     // 64x256, 8 waves, 2 workgroups / CU
            case 96: return launch_cfg<KH, KW, 1, 2, 3, 4, KC>(p, st);     // 96x256, 12 waves
            default: return launch_cfg<KH, KW, 1, 2, 4, 4, KC>(p, st);     // 128x256, 16 waves
        }
    }
    switch (mb) {
        case 32: return launch_cfg<KH, KW, 1, 2, 1, 4, KC>(p, st);
        case 64:
            if (p->T >= 64) return launch_cfg<KH, KW, 1, 2, 2, 8, KC>(p, st);     // 16 waves, 64x512
            return launch_cfg<KH, KW, 2, 2, 1, 4, KC>(p, st);
        case 96: return launch_cfg<KH, KW, 1, 2, 3, 4, KC>(p, st);     // 12 waves, 96x256
        default: return launch_cfg<KH, KW, 1, 2, 4, 4, KC>(p, st);     // 16 waves (4 per SIMD), 128x256
    }
}

int aid_conv53_dma_try(const aid_conv2d_params* p, hipStream_t st);   // aid_conv_dma.hip
int aid_conv53_wino_try(const aid_conv2d_params* p, hipStream_t st);  // aid_conv_wino.hip
int aid_conv53_wino2d(const aid_conv2d_params* p, hipStream_t st);    // aid_wino2d.hip
int aid_conv1x1_stream_try(const aid_conv2d_params* p, hipStream_t st);  // aid_conv1x1.hip
int aid_conv1x1_dma_try(const aid_conv2d_params* p, hipStream_t st);     // aid_conv1x1_dma.hip
int aid_conv_small_try(const aid_conv2d_params* p, hipStream_t st);      // aid_conv_small.hip
int aid_conv1x1_rs_try(const aid_conv2d_params* p, hipStream_t st);      // aid_conv1x1_rs.hip

extern "C" int aid_conv2d(const aid_conv2d_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->x.p && p->y.p && p->wp, "aid_conv2d: null pointer");
    AID_REQUIRE(p->B > 0 && p->Cin > 0 && p->Cout > 0 && p->F > 0 && p->T > 0, "aid_conv2d: empty shape");
    AID_REQUIRE((p->T % 4) == 0, "aid_conv2d: T must be a multiple of 4");
    AID_REQUIRE(p->x_wino == 3 || p->x_wino == 4 || ((p->x.sB % 4) == 0 && (p->x.sC % 4) == 0 && (p->x.sF % 4) == 0), "aid_conv2d: input view strides % 4 == 0");
    AID_REQUIRE((((uintptr_t)p->x.p) & 15) == 0, "aid_conv2d: input view must be 16-byte aligned");
    int cip, cop;
    aid_conv2d_pack_dims(p->Cin, p->Cout, &cip, &cop);
    AID_REQUIRE(p->Cin_pad == cip && p->Cout_pad == cop, "aid_conv2d: packed weight dims mismatch (use aid_conv2d_pack_dims)");
    AID_REQUIRE(p->epi == 0 || (p->epi == 1 && p->aux.p && p->aux_scale), "aid_conv2d: epi=1 needs aux + aux_scale");
    AID_REQUIRE((int64_t)p->B * p->in_scale_ld < (1LL << 31), "aid_conv2d: in_scale too large");
    AID_REQUIRE(!p->x_wino || (p->KH == 5 && p->KW == 3 && p->wp_wino), "aid_conv2d: x_wino is a 5x3 Winograd-path input layout");
    AID_REQUIRE(!p->fin_mode || (p->x_wino && p->KH == 5 && p->KW == 3), "aid_conv2d: fin_mode is an option of the row-shared 5x3 Winograd kernels (aid_conv2d_fin_supported)");
    if (p->x_wino == 3 || p->x_wino == 4) return aid_conv53_wino2d(p, st);              // non-fused 2-D Winograd form: batched GEMM + output pass (aid_wino2d.hip)
    if (p->x2.p) {
        AID_REQUIRE(p->KH == 1 && p->KW == 1 && !p->in_scale && p->act == 0 && p->Cin1 > 0 && p->Cin1 < p->Cin && (p->Cin1 % 16) == 0 && ((p->Cin - p->Cin1) % 16) == 0,
                    "aid_conv2d: x2 is an option of plain 1x1 convolutions with both K segments multiples of 16");
        int r = aid_conv1x1_rs_try(p, st);
        if (r == 0) r = aid_conv1x1_dma_try(p, st);
        if (r == 0) aid_set_error("aid_conv2d: x2 given but the layer is not eligible for the 1x1 kernels that take it");
        return r == 1 ? AID_OK : (r < 0 ? r : AID_E_BADARG);
    }
    if (p->stat_ws) {
        AID_REQUIRE(!p->dot_ws && p->epi == 0 && p->x_wino, "aid_conv2d: stat_ws is an option of the forward epilogue on Winograd-domain input");
        AID_REQUIRE(p->stat_n > 0 && p->stat_n == aid_conv2d_stat_partials(p->B, p->Cin, p->Cout, p->F, p->T, p->dilF, p->x_wino), "aid_conv2d: stat_n != aid_conv2d_stat_partials()");
    }
    if (p->dot_ws && p->KH == 1 && p->KW == 1) {                     // <y, aux> partials from the direct-to-LDS 1x1 kernel's dGELU epilogue
        AID_REQUIRE(p->epi == 1 && !p->res.p, "aid_conv2d: dot_ws on a 1x1 layer is an option of the dGELU epilogue without residual");
        int r = aid_conv1x1_rs_try(p, st);
        if (r == 0) r = aid_conv1x1_dma_try(p, st);
        if (r == 0) aid_set_error("aid_conv2d: dot_ws given but the 1x1 layer is not eligible (aid_conv2d_dot_partials_1x1)");
        return r == 1 ? AID_OK : (r < 0 ? r : AID_E_BADARG);
    }
    if (p->dot_ws) {
        AID_REQUIRE(p->KH == 5 && p->KW == 3 && p->wp_wino && p->wino_taps == (p->x_wino == 2 ? 50 : 30) && p->epi == 1 && !p->res.p,
                    "aid_conv2d: dot_ws is an option of the F(4,3) / F(8,3) dGELU epilogue");
        AID_REQUIRE(p->dot_n > 0 && p->dot_n == aid_conv2d_dot_partials(p->B, p->Cin, p->Cout, p->F, p->T, p->dilF, p->x_wino), "aid_conv2d: dot_n != aid_conv2d_dot_partials()");
        const int r = aid_conv53_wino_try(p, st);
        if (r == 0) aid_set_error("aid_conv2d: dot_ws given but the layer is not eligible for the F(4,3) kernels");
        return r == 1 ? AID_OK : (r < 0 ? r : AID_E_BADARG);
    }
    if (p->x_wino) {
        const int r = aid_conv53_wino_try(p, st);
        return r == 1 ? AID_OK : (r < 0 ? r : AID_E_BADARG);
    }
    if (p->Cin <= 8 || p->Cout <= 8) {
        const int r = aid_conv_small_try(p, st);             // 2- / 8-channel projections: VALU streaming kernels
        if (r != 0) return r < 0 ? r : AID_OK;
    }
    if (p->KH == 5 && p->KW == 3) {
        if (p->wp_wino) {
            const int r = aid_conv53_wino_try(p, st);    // Winograd F(4,3) along T: 6 MFMAs per 4 outputs instead of 12
            if (r != 0) return r < 0 ? r : AID_OK;
        }
        {
            const int r = aid_conv53_dma_try(p, st);     // direct form, direct-to-LDS staging (layers without a Winograd pack)
            if (r != 0) return r < 0 ? r : AID_OK;
        }
        return launch_m<5, 3, 4>(p, st);                 // register-staged kernel: in-kernel prologue, Cin = 2, odd shapes
    }
    if (p->KH == 1 && p->KW == 1) {
        {
            int r = aid_conv1x1_rs_try(p, st);               // K <= 256, HBM-bound: register-streamed kernel (weights resident in LDS)
            if (r != 0) return r < 0 ? r : AID_OK;
            r = aid_conv1x1_dma_try(p, st);                  // direct-to-LDS kernel, 2-3 workgroups per CU (K % 16 == 0, Cout tile 64/96/128)
            if (r != 0) return r < 0 ? r : AID_OK;
            r = aid_conv1x1_stream_try(p, st);               // remaining short-K / narrow projections: streaming kernel
            if (r != 0) return r < 0 ? r : AID_OK;
        }
        if (p->Cin <= 8) return launch_m<1, 1, 8>(p, st);
        return launch_m<1, 1, 16>(p, st);
    }
    aid_set_error("aid_conv2d: unsupported kernel size (5x3 and 1x1 only)");
    return AID_E_BADARG;
}
