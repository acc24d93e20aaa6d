// 1x1 channel projections (res_conv / proj_in / proj_out / skip projections, reference unet...py:412-415,:488-491) as a
// STREAMING fp32-MFMA GEMM.  These layers have K = Cin of 64..512 only: 20-90 FLOP per byte, i.e. at or below the
// HBM ridge of the fp32 matrix pipe, so the kernel is built around memory traffic, not around the matrix core:
//
//   * one wave owns 32*NT consecutive output positions of one (b,f) row segment and ALL 32*MT output channels of its
//     Cout slice: every activation is read from HBM exactly once per Cout slice, straight from global memory into the
//     register that feeds the MFMA -- no LDS staging, no barrier, no inter-wave dependence, so many independent waves
//     per CU keep loads in flight;
//   * lane (n = lane & 31, k = lane >> 5) loads NT consecutive samples x[b, ci0+k, f, t0 + NT*n ...] with one
//     dwordxNT load.  Register r of that vector, taken across the 32 lanes, is the B operand of "tile r": the positions
//     {NT*n + r}.  An MFMA does not care which positions form its N dimension, and in the epilogue the NT tiles of a
//     lane are again NT consecutive samples -> dwordxNT stores;
//   * the A operand (weights, [Cin_pad][Cout_pad], cout contiguous, <= 1 MB: L2 resident) is a coalesced dword load per
//     (k-step, m-tile), scaled by in_scale[b,ci] when the layer has a per-(b,ci) prologue scale;
//   * loads run PF k-steps ahead of the MFMAs that consume them (register ring).
// Epilogue identical to conv_mfma_kernel: y = alpha*(res_scale*res + acc*out_scale[b,co]) or the dGELU form.
#include "aid_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct C1Dev {
    aid_conv2d_params p;
    int64_t npos;          // B*F*T
    int mchunks;           // Cout_pad / (32*MT)
};

template <int NT> struct VecT;
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };

template <int NT> __device__ __forceinline__ float vget(const typename VecT<NT>::type& v, int r);
template <> __device__ __forceinline__ float vget<2>(const float2& v, int r) { return r == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float vget<4>(const float4& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

template <int MT, int NT, int PF, int MINW>
__global__ __launch_bounds__(256, MINW) void conv1x1_stream_kernel(const C1Dev a) {
    typedef typename VecT<NT>::type vec_t;
    const aid_conv2d_params& p = a.p;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    // workgroup id -> (position block, Cout slice): Cout slice fastest so the slices of the same activations run together
    const int mc = blockIdx.x % a.mchunks;
    const int64_t pb = blockIdx.x / a.mchunks;
    const int64_t P0 = (pb * 4 + wave) * (32 * NT);
    if (P0 >= a.npos) return;
    const int64_t pos = P0 + NT * (lane & 31);
    const int64_t row = pos / p.T;
    const int t = (int)(pos - row * p.T);
    const int b = (int)(row / p.F);
    const int f = (int)(row - (int64_t)b * p.F);
    const int m0 = mc * (32 * MT);

    const float* xp = p.x.p + (int64_t)b * p.x.sB + (int64_t)f * p.x.sF + t + (int64_t)half * p.x.sC;
    const int64_t xstep = 2 * p.x.sC;
    const float* wp = p.wp + (int64_t)half * p.Cout_pad + m0 + (lane & 31);
    const int64_t wstep = 2 * (int64_t)p.Cout_pad;
    const float* sp = p.in_scale ? p.in_scale + (int64_t)b * p.in_scale_ld + half : nullptr;
    const int nsteps = p.Cin >> 1;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec_t bq[PF];
    float aq[PF][MT];
    float sq[PF];
    auto load = [&](int s, int slot) {
        bq[slot] = *reinterpret_cast<const vec_t*>(xp + (int64_t)s * xstep);
#pragma unroll
        for (int i = 0; i < MT; ++i) aq[slot][i] = wp[(int64_t)s * wstep + 32 * i];
        sq[slot] = sp ? sp[2 * s] : 1.f;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (d < nsteps) load(d, d);
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int s = s0 + d;
            if (s < nsteps) {
                const vec_t bv = bq[d];
                float av[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = aq[d][i] * sq[d];
                if (s + PF < nsteps) load(s + PF, d);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], vget<NT>(bv, j), acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: NT consecutive samples per lane and output channel ------------------------------------------------
    const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
    const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
    const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int mbase = m0 + i * 32 + 4 * half;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {             // gather 4 rows, then compute + store (res may alias y)
            vec_t rv[4], uv[4];
            float sv[4], as[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                const bool ok = m < p.Cout;
                if (ok && p.res.p) rv[q] = *reinterpret_cast<const vec_t*>(p.res.p + rbase + (int64_t)m * p.res.sC);
                sv[q] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                if (ok && p.epi == 1) {
                    as[q] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                    uv[q] = *reinterpret_cast<const vec_t*>(p.aux.p + abase + (int64_t)m * p.aux.sC);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m >= p.Cout) continue;
                float o[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float v = acc[i][j][r] * sv[q];
                    if (p.epi == 1) v *= aid_dgelu(vget<NT>(uv[q], j) * as[q]);
                    if (p.res.p) v += p.res_scale * vget<NT>(rv[q], j);
                    o[j] = p.alpha * v;
                }
                vec_t ov;
                if constexpr (NT == 2) ov = make_float2(o[0], o[1]);
                else ov = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<vec_t*>(p.y.p + ybase + (int64_t)m * p.y.sC) = ov;
            }
        }
    }
}

template <int MT, int NT, int PF, int MINW>
static int launch_c1(const aid_conv2d_params* p, hipStream_t st) {
    C1Dev a;
    a.p = *p;
    a.npos = (int64_t)p->B * p->F * p->T;
    a.mchunks = p->Cout_pad / (32 * MT);
    const int64_t pblocks = (a.npos + 4 * 32 * NT - 1) / (4 * 32 * NT);
    hipLaunchKernelGGL((conv1x1_stream_kernel<MT, NT, PF, MINW>), dim3((unsigned)(pblocks * a.mchunks)), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv1x1_stream_kernel");
    return AID_OK;
}

static bool view_ok(const aid_view& v, int al) {
    return (v.sB % al) == 0 && (v.sC % al) == 0 && (v.sF % al) == 0 && (((uintptr_t)v.p) & (4 * al - 1)) == 0;
}

// returns 1 if the streaming kernel took the launch, 0 if not eligible, <0 on error
int aid_conv1x1_stream_try(const aid_conv2d_params* p, hipStream_t st) {
    if (!(p->KH == 1 && p->KW == 1) || p->act != 0) return 0;
    if (p->Cin < 16 || (p->Cin & 1) || p->Cout < 32) return 0;
    const int64_t ft = (int64_t)p->F * p->T;
    if (p->F == 1 || (int64_t)p->B * ft < 2048) return 0;            // few positions (qk GEMMs, F = 1): the tiled kernel's job
    // measured (profiles/r01_conv1x1_probe.txt): ahead of the register-staged tiled kernel for the short-K / write-dominated
    // projections (Cin <= 96: 8-35 %), behind it (10-18 %) once K >= 256 makes the layer MFMA-bound; the direct-to-LDS kernel
    // (aid_conv1x1_dma.hip, tried first) is 17-20 % ahead of this one wherever it is eligible, so this kernel serves the shapes
    // that one refuses (K % 16 != 0, Cout tiles of 32).  128 positions per wave (float4 traffic) when the geometry allows, else 64.
    if (!(p->Cin <= 96 || p->Cout_pad <= 32)) return 0;
    int nt = 4;
    if ((ft % 128) || (p->T % 4)) nt = 2;
    if ((ft % (32 * nt)) || (p->T % nt)) return 0;                  // a wave's positions stay inside one sample b, vectors inside one row
    if (!view_ok(p->x, nt) || !view_ok(p->y, nt)) return 0;
    if (p->res.p && !view_ok(p->res, nt)) return 0;
    if (p->aux.p && !view_ok(p->aux, nt)) return 0;
    const int mt32 = p->Cout_pad / 32;
    int rc;
    if (nt == 4) {
        if (mt32 % 2 == 0) rc = launch_c1<2, 4, 4, 2>(p, st);
        else               rc = launch_c1<1, 4, 4, 2>(p, st);
    } else {
        if (mt32 % 4 == 0)      rc = launch_c1<4, 2, 4, 2>(p, st);   // 128-wide slices, 2 waves / SIMD
        else if (mt32 % 3 == 0) rc = launch_c1<3, 2, 4, 2>(p, st);   // 96-wide (Cout 96, 192), 3 waves / SIMD
        else if (mt32 % 2 == 0) rc = launch_c1<2, 2, 4, 2>(p, st);
        else                    rc = launch_c1<1, 2, 4, 2>(p, st);
    }
    return rc == AID_OK ? 1 : rc;
}
