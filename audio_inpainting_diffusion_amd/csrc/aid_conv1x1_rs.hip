// conv11_rs_kernel: 1x1 channel projections with K <= 256 as a REGISTER-STREAMED fp32-MFMA kernel (round 3).
//
// These layers (res_conv / proj / pyramid-path 1x1 of unet...py:412-415, :488-491, :700-712 and their input gradients) move 16-60 FLOP per
// byte: below the ridge by the byte count.  The LDS-staged tile kernels (conv11_dma_kernel, conv_mfma_kernel) reach 1.8-3.3 TB/s on them: a
// 64..128 x 256 tile lives ~35 us (four 16-channel chunks, each a direct-to-LDS burst + workgroup barrier, then an epilogue whose residual / aux
// loads are issued after the last MFMA), with two or three tiles per CU in flight.  This kernel reaches 2.5-3.7 TB/s; its ablation
// (an experimental build with switches that drop the stores / the loads / the MFMAs: profiles/r03_c11_probe.txt) shows what is left: with fp32 32x32x2 MFMAs the matrix time of these layers is
// 60-90 % of their HBM time, the dGELU epilogue adds a VALU term of the same order, and a wave overlaps the three only against the other wave
// of its SIMD.  Here
//   * a workgroup (8 waves) loads the layer's WHOLE weight matrix slice [K x 32 MT] into LDS once and then streams position tiles past it;
//   * every wave is its own pipeline over tiles of 64 positions x all 32 MT output channels: the B operand (activations) goes HBM -> REGISTERS
//     (float2 per lane = the two interleaved position tiles, 2 x 256 contiguous bytes per load), U = 16 k-steps = 8 KB in flight per wave,
//     64 KB per CU; no LDS traffic for activations, no workgroup barrier after the weight load;
//   * the per-(b, ci) prologue scale and the GELU prologue (act = 1) are applied to the B values in registers (each activation is loaded by
//     exactly one wave per Cout slice, so nothing is recomputed within a slice);
//   * the epilogue (out_scale, dGELU(aux), residual, alpha, <y, aux> partials) is conv11_dma_kernel's, per wave.
// The k-pairs are accumulated in increasing order into one accumulator per output, as in the other 1x1 kernels.
#include "aid_common.h"
#include <stdint.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct C11RDev {
    aid_conv2d_params p;
    int nsteps;                // k-pairs (Cin / 2)
    int nsteps1;               // k-pairs served by p.x (all unless p.x2 is given)
    int tps;                   // 64-position tiles per sample (F*T / 64)
    int ntiles;                // B * tps
};

// returns the number of <y, aux> partial slots per (sample, group) = tiles per sample, or 0 when the shape is not served
// Which layers (a function of the per-sample shape only, so that a segment's result does not depend on its batch): measured per shape at batch 8 AND
// batch 1 against the tile kernels (profiles/r03_trace_b8.txt / _b1.txt before and after): 96 output channels (MT = 3) win for every K <= 256
// (batch 8: -12 ... -32 %); 64 output channels (MT = 2) win at K = 64 and lose at K = 128 / 192 (more waves would be needed to cover the longer K
// loop's exposed first block); 128-channel slices (MT = 4: 128 accumulators + the epilogue's gather registers, 64-128 KB of weights per workgroup)
// lose on the small pyramid levels by 2x (32 ... 128 tiles per sample: one wave per tile, most SIMDs idle) and are not instantiated.
static int c11rs_shape(int Cin, int Cout, int cop, int F, int T) {
    if (F <= 1 || Cin < 32 || (Cin % 32) || Cin > 256 || Cout < 32) return 0;
    if (!(cop == 96 || (cop == 64 && Cin <= 64)) || Cout != cop) return 0;
    if ((T % 2) || (((int64_t)F * T) % 64)) return 0;
    return (int)(((int64_t)F * T) / 64);
}

// EPI: the dGELU(aux) epilogue (epi = 1); RES: a residual is added.  Cout == Cout_pad (host-checked): no row predicates in the epilogue.
template <int MT, int KMAX, bool ACT, bool SCALE, bool EPI, bool RES>
__global__ __launch_bounds__(512, 1) void conv11_rs_kernel(const C11RDev a) {
    constexpr int U = 16;                                // k-steps in flight per wave
    constexpr int WROW = 32 * MT;
    const aid_conv2d_params& p = a.p;
    __shared__ __attribute__((aligned(16))) float wl[KMAX * WROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int n = lane & 31;
    const int m0 = blockIdx.y * WROW;                    // Cout slice of this workgroup

    // ---- weights of this Cout slice: [Cin][32 MT] -> LDS, once ------------------------------------------------------------------
    {
        const int nvec = p.Cin * (WROW / 4);
        for (int e = tid; e < nvec; e += 512) {
            const int ci = e / (WROW / 4), c4 = e % (WROW / 4);
            *reinterpret_cast<float4*>(wl + ci * WROW + 4 * c4) = *reinterpret_cast<const float4*>(p.wp + (int64_t)ci * p.Cout_pad + m0 + 4 * c4);
        }
    }
    __syncthreads();

    // Addressing: per-sample base pointers are wave-uniform (scalar registers); a lane's offsets within the sample are 32-bit byte offsets
    // that advance by adding a scalar stride -- the 64-bit multiply-adds of the straightforward form were a third of the kernel's VALU issue.
    // The context of a tile (sample, row, first position, load addresses) is set by `seek`.  Measured and dropped: requesting the first U k-steps of the
    // NEXT tile before the epilogue of the current one (the ring stays live through the epilogue: 96-channel dGELU instances spill; the others
    // -1 ... +8 % per launch, K > 64 on the losing side, end to end +0.6 % instead of +0.9 %).
    const int wstep = gridDim.x * 8;
    const unsigned xs = 8u * (unsigned)p.x.sC, xs2 = 8u * (unsigned)p.x2.sC;
    int b = 0, tin = 0, f = 0, t = 0;
    const char* xb = nullptr;
    const char* xb2 = nullptr;
    unsigned xo = 0, xo2 = 0;
    const float* sp = nullptr;
    float2 ring[U];
    float scr[SCALE ? U : 1];
    auto seek = [&](int tl) {
        b = tl / a.tps;
        tin = tl - b * a.tps;
        const int pos = tin * 64 + 2 * n;                 // first of this lane's two positions within the sample
        f = pos / p.T; t = pos - f * p.T;
        xb = reinterpret_cast<const char*>(p.x.p + (int64_t)b * p.x.sB);
        xb2 = p.x2.p ? reinterpret_cast<const char*>(p.x2.p + (int64_t)b * p.x2.sB) : xb;
        xo = 4u * (unsigned)(f * (int)p.x.sF + t + half * (int)p.x.sC);
        xo2 = p.x2.p ? 4u * (unsigned)(f * (int)p.x2.sF + t + half * (int)p.x2.sC) : xo;
        sp = SCALE ? p.in_scale + (int64_t)b * p.in_scale_ld + half : nullptr;
    };
    auto ld = [&](int s, int q) {                         // s is wave-uniform
        if (s < a.nsteps1) ring[q] = *reinterpret_cast<const float2*>(xb + (xo + (unsigned)s * xs));
        else ring[q] = *reinterpret_cast<const float2*>(xb2 + (xo2 + (unsigned)(s - a.nsteps1) * xs2));
        if (SCALE) scr[q] = sp[2 * s];
    };
    for (int tl = blockIdx.x * 8 + wave; tl < a.ntiles; tl += wstep) {
        seek(tl);
#pragma unroll
        for (int q = 0; q < U; ++q) ld(q, q);
        f32x16 acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int s0 = 0; s0 < a.nsteps; s0 += U) {
            const bool more = s0 + U < a.nsteps;
            const float* wr0 = wl + (2 * s0 + half) * WROW + n;
#pragma unroll
            for (int q = 0; q < U; ++q) {
                float bx = ring[q].x, by = ring[q].y;
                if (SCALE) { bx *= scr[q]; by *= scr[q]; }
                if (more) ld(s0 + U + q, q);
                if (ACT) { bx = aid_gelu(bx); by = aid_gelu(by); }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float aw = wr0[2 * q * WROW + 32 * i];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bx, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, by, acc[i][1], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: two consecutive positions per lane and output channel ------------------------------------------------------
        const int be = b, tine = tin;
        char* yb = reinterpret_cast<char*>(p.y.p + (int64_t)b * p.y.sB);
        const char* rb = RES ? reinterpret_cast<const char*>(p.res.p + (int64_t)b * p.res.sB) : nullptr;
        const char* ab = EPI ? reinterpret_cast<const char*>(p.aux.p + (int64_t)b * p.aux.sB) : nullptr;
        const float* osb = p.out_scale ? p.out_scale + (int64_t)b * p.out_scale_ld : nullptr;
        const float* asb = EPI ? p.aux_scale + (int64_t)b * p.aux_scale_ld : nullptr;
        const unsigned ypos = 4u * (unsigned)(f * (int)p.y.sF + t);
        const unsigned rpos = RES ? 4u * (unsigned)(f * (int)p.res.sF + t) : 0u;
        const unsigned upos = EPI ? 4u * (unsigned)(f * (int)p.aux.sF + t) : 0u;
        float dsum[MT][4];                                // <y, aux> per block of 4 channels (dot_ws)
        int hv = half;
        asm volatile("" : "+v"(hv));                      // (opaque: the row offsets m * sC are tile-invariant and would otherwise be hoisted out of the
                                                          //  tile loop into ~190 registers that spill)
        const unsigned ys = 4u * (unsigned)p.y.sC, rs = 4u * (unsigned)p.res.sC, us = 4u * (unsigned)p.aux.sC;
        const int mb0 = m0 + 4 * hv;
        const unsigned yo = ypos + (unsigned)mb0 * ys;
        const unsigned ro = RES ? rpos + (unsigned)mb0 * rs : 0u;
        const unsigned uo = EPI ? upos + (unsigned)mb0 * us : 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = mb0 + 32 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) dsum[i][q] = 0.f;
            constexpr int RB = 8;                         // rows gathered per batch
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RB) {         // gather RB rows, then compute + store (res may alias y)
                float2 rv[RES ? RB : 1], uv[EPI ? RB : 1];
                float sv[RB], as[EPI ? RB : 1];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const int dm = (r & 3) + 8 * (r >> 2);                 // row within the 32-channel tile (+ 4 half)
                    if (RES) rv[q] = *reinterpret_cast<const float2*>(rb + (ro + (unsigned)(32 * i + dm) * rs));
                    sv[q] = osb ? osb[mbase + dm] : 1.f;
                    if (EPI) {
                        as[q] = asb[mbase + dm];
                        uv[q] = *reinterpret_cast<const float2*>(ab + (uo + (unsigned)(32 * i + dm) * us));
                    }
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const int dm = (r & 3) + 8 * (r >> 2);
                    float y0 = acc[i][0][r] * sv[q], y1 = acc[i][1][r] * sv[q];
                    if (EPI) { y0 *= aid_dgelu(uv[q].x * as[q]); y1 *= aid_dgelu(uv[q].y * as[q]); }
                    if (RES) { y0 += p.res_scale * rv[q].x; y1 += p.res_scale * rv[q].y; }
                    y0 *= p.alpha; y1 *= p.alpha;
                    *reinterpret_cast<float2*>(yb + (yo + (unsigned)(32 * i + dm) * ys)) = make_float2(y0, y1);
                    if (EPI) dsum[i][r >> 2] += y0 * uv[q].x + y1 * uv[q].y;
                }
            }
        }
        if (EPI && p.dot_ws) {                            // per (sample, channel group): one partial per 64-position tile, fixed order
            const int cpg = p.Cout >> 3;                  // channels per group (8 groups); cpg % 4 == 0 and (32 MT) % cpg == 0 (host-checked)
            const int g = m0 / cpg + lane;                // lanes 0 .. (32 MT / cpg) - 1 own one group each
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = dsum[i][q];
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);
                    const float w = __shfl_xor(v, 32);    // the other half's block (channels + 4)
                    const float v0 = half ? w : v, v1 = half ? v : w;
                    const int c0 = m0 + 32 * i + 8 * q;   // channels c0 .. c0+3 (half 0), c0+4 .. c0+7 (half 1)
                    if (c0 / cpg == g) sacc += v0;
                    if ((c0 + 4) / cpg == g) sacc += v1;
                }
            if (lane < WROW / cpg && g < 8) p.dot_ws[((int64_t)be * 8 + g) * p.dot_n + tine] = (double)sacc;
        }
    }
}

template <int MT, int KMAX>
static int launch_c11rs(const aid_conv2d_params* p, hipStream_t st) {
    C11RDev a;
    a.p = *p;
    a.nsteps = p->Cin / 2;
    a.nsteps1 = p->x2.p ? p->Cin1 / 2 : a.nsteps;
    a.tps = (int)(((int64_t)p->F * p->T) / 64);
    a.ntiles = p->B * a.tps;
    int gx = aid_cdiv(a.ntiles, 8);
    const int ny = p->Cout_pad / (32 * MT);
    const int cap = ny > 1 ? 256 / ny + (256 % ny ? 1 : 0) : 256;      // one workgroup per CU over all Cout slices
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)ny);
#define AID_C11RS(ACTv, SCv, EPIv, RESv) hipLaunchKernelGGL((conv11_rs_kernel<MT, KMAX, ACTv, SCv, EPIv, RESv>), grid, dim3(512), 0, st, a)
    const int key = (p->act ? 8 : 0) | (p->in_scale ? 4 : 0) | (p->epi == 1 ? 2 : 0) | (p->res.p ? 1 : 0);
    switch (key) {                                       // (the combinations the network emits; anything else goes to the tile kernels)
        case 12: AID_C11RS(true, true, false, false); break;
        case 13: AID_C11RS(true, true, false, true); break;
        case 0:  AID_C11RS(false, false, false, false); break;
        case 1:  AID_C11RS(false, false, false, true); break;
        case 2:  AID_C11RS(false, false, true, false); break;
        case 3:  AID_C11RS(false, false, true, true); break;
        case 4:  AID_C11RS(false, true, false, false); break;
        case 5:  AID_C11RS(false, true, false, true); break;
        case 6:  AID_C11RS(false, true, true, false); break;
        default: return 1000;
    }
#undef AID_C11RS
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv11_rs_kernel");
    return AID_OK;
}

// <y, aux> partial slots per (sample, group) when this kernel serves the shape (0: it does not)
int aid_conv1x1_rs_dot_partials(int Cin, int Cout, int cop, int F, int T) {
    const int n = c11rs_shape(Cin, Cout, cop, F, T);
    if (!n || (Cout % 8)) return 0;
    const int cpg = Cout / 8;
    const int mblk = (cop % 128 == 0) ? 128 : cop;
    if ((cpg % 4) || (mblk % cpg)) return 0;
    return n;
}

int aid_conv1x1_rs_shape_ok(int Cin, int Cout, int cop, int F, int T) { return c11rs_shape(Cin, Cout, cop, F, T) ? 1 : 0; }

int aid_conv1x1_rs_enabled(void) {
#ifdef AID_EXPERIMENT
    static const int enabled = getenv("AID_C11_RS") ? atoi(getenv("AID_C11_RS")) : 1;
#else
    constexpr int enabled = 1;
#endif
    return enabled;
}

// returns 1 if this kernel took the launch, 0 if not eligible, <0 on error
int aid_conv1x1_rs_try(const aid_conv2d_params* p, hipStream_t st) {
    if (!aid_conv1x1_rs_enabled()) return 0;
    if (!(p->KH == 1 && p->KW == 1) || p->ws || (p->act && !p->in_scale) || p->Cout != p->Cout_pad) return 0;
    if (!c11rs_shape(p->Cin, p->Cout, p->Cout_pad, p->F, p->T)) return 0;
    auto al = [](const aid_view& v, int q) { return (v.sB % q) == 0 && (v.sC % q) == 0 && (v.sF % q) == 0 && (((uintptr_t)v.p) & (4 * q - 1)) == 0; };
    if (!al(p->x, 2) || !al(p->y, 2) || (p->res.p && !al(p->res, 2)) || (p->aux.p && !al(p->aux, 2))) return 0;
    if (p->x2.p && (!al(p->x2, 2) || (p->Cin1 % 2) || p->Cin1 <= 0 || p->Cin1 >= p->Cin || p->in_scale || p->act)) return 0;
    if ((((uintptr_t)p->wp) & 15) || (p->Cout_pad % 4)) return 0;
    auto fits = [&](const aid_view& v, int C) { return v.sC >= 0 && v.sF >= 0 && (int64_t)4 * ((int64_t)C * v.sC + (int64_t)p->F * v.sF + p->T) < (1LL << 32); };   // 32-bit byte offsets within a sample
    if (!fits(p->x, p->Cin) || !fits(p->y, p->Cout_pad) || (p->res.p && !fits(p->res, p->Cout_pad)) || (p->aux.p && !fits(p->aux, p->Cout_pad)) || (p->x2.p && !fits(p->x2, p->Cin))) return 0;
    if (p->dot_ws && (p->epi != 1 || p->dot_n <= 0 || p->dot_n != aid_conv1x1_rs_dot_partials(p->Cin, p->Cout, p->Cout_pad, p->F, p->T))) return 0;
    int rc;
    const bool k128 = p->Cin <= 128;
    if (p->Cout_pad == 96) rc = k128 ? launch_c11rs<3, 128>(p, st) : launch_c11rs<3, 256>(p, st);
    else                   rc = launch_c11rs<2, 128>(p, st);
    if (rc == 1000) return 0;
    return rc == AID_OK ? 1 : rc;
}
