// Few-channel convolutions on the VALU: the layers with 2 or 8 channels on one side (pyramid projections 2 -> C 5x3 and
// their input gradients C -> 2, out-block projections C -> 2, attention proj_in / proj_out C <-> 8; reference
// unet...py:412-415, :676, :719).  On the MFMA kernels a 32-row Cout tile (or a 2-channel K chunk) is 94 % padding and the
// layer runs at 0.3-3 TB/s; these are pure streaming problems:
//   small-Cout kernel: one thread owns 4 consecutive samples of one (b,f) row and ALL (<= 8) output channels, walks the
//                      input channels once (float4 row loads + the two neighbours for the 3-tap rows), weights come through
//                      scalar loads (uniform index);
//   small-Cin  kernel: one thread keeps its (<= 8 channels x KH rows x 6 samples) input patch in registers and walks the
//                      output channels, one float4 store each.
// Same prologue / epilogue contract as aid_conv2d (in_scale, out_scale, res, alpha); act = 0 and epi = 0 only.
#include "aid_common.h"
#include <stdlib.h>

struct CsDev { aid_conv2d_params p; int lpr_log2, nrows, tiles; };

template <int KH, int KW>
__device__ __forceinline__ void load_patch(const aid_conv2d_params& p, const float* xr, int f, int t4, float (&v)[KH][4 + (KW > 1 ? 2 : 0)]) {
    // v[kh][0..5] = x[f + (kh-KH/2)*dil][t4-1 .. t4+4] for KW = 3 (zero outside), v[kh][0..3] = x[..][t4 .. t4+3] for KW = 1
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
        const int fi = f + (kh - KH / 2) * p.dilF;
        const bool ok = fi >= 0 && fi < p.F;
        const float* r = xr + (int64_t)fi * p.x.sF;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) c = *reinterpret_cast<const float4*>(r + t4);
        if (KW > 1) {
            v[kh][0] = (ok && t4 > 0) ? r[t4 - 1] : 0.f;
            v[kh][1] = c.x; v[kh][2] = c.y; v[kh][3] = c.z; v[kh][4] = c.w;
            v[kh][5] = (ok && t4 + 4 < p.T) ? r[t4 + 4] : 0.f;
        } else {
            v[kh][0] = c.x; v[kh][1] = c.y; v[kh][2] = c.z; v[kh][3] = c.w;
        }
    }
}

__device__ __forceinline__ bool cs_index(const CsDev& a, int& b, int& f, int& t4) {
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    if (row >= a.nrows) return false;
    t4 = (tile * lpr + lq) * 4;
    if (t4 >= a.p.T) return false;
    b = row / a.p.F;
    f = row - b * a.p.F;
    return true;
}

// ---- Cout <= CO (2 or 8), any Cin -------------------------------------------------------------------------------------
template <int KH, int KW, int CO>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const CsDev a) {
    const aid_conv2d_params& p = a.p;
    int b, f, t4;
    if (!cs_index(a, b, f, t4)) return;
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) { acc[co][0] = 0.f; acc[co][1] = 0.f; acc[co][2] = 0.f; acc[co][3] = 0.f; }
    const float* xb = p.x.p + (int64_t)b * p.x.sB;
    const float* sp = p.in_scale ? p.in_scale + (int64_t)b * p.in_scale_ld : nullptr;
    for (int ci = 0; ci < p.Cin; ++ci) {
        float v[KH][4 + (KW > 1 ? 2 : 0)];
        load_patch<KH, KW>(p, xb + (int64_t)ci * p.x.sC, f, t4, v);
        const float s = sp ? sp[ci] : 1.f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const float* w = p.wp + ((int64_t)(kh * KW + kw) * p.Cin_pad + ci) * p.Cout_pad;     // uniform: scalar loads
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = w[co] * s;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co][e] += wv * v[kh][e + kw];
                }
            }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co >= p.Cout) break;
        const float os = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + co] : 1.f;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res.p) r = *reinterpret_cast<const float4*>(p.res.p + (int64_t)b * p.res.sB + (int64_t)co * p.res.sC + (int64_t)f * p.res.sF + t4);
        float4 o;
        o.x = p.alpha * (p.res_scale * r.x + acc[co][0] * os); o.y = p.alpha * (p.res_scale * r.y + acc[co][1] * os);
        o.z = p.alpha * (p.res_scale * r.z + acc[co][2] * os); o.w = p.alpha * (p.res_scale * r.w + acc[co][3] * os);
        *reinterpret_cast<float4*>(p.y.p + (int64_t)b * p.y.sB + (int64_t)co * p.y.sC + (int64_t)f * p.y.sF + t4) = o;
    }
}

// ---- Cin <= CI (2 or 8), any Cout ---------------------------------------------------------------------------------------
template <int KH, int KW, int CI>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const CsDev a) {
    const aid_conv2d_params& p = a.p;
    int b, f, t4;
    if (!cs_index(a, b, f, t4)) return;
    float v[CI][KH][4 + (KW > 1 ? 2 : 0)];
    const float* xb = p.x.p + (int64_t)b * p.x.sB;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        if (ci < p.Cin) {
            load_patch<KH, KW>(p, xb + (int64_t)ci * p.x.sC, f, t4, v[ci]);
            const float s = p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_ld + ci] : 1.f;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int e = 0; e < 4 + (KW > 1 ? 2 : 0); ++e) v[ci][kh][e] *= s;
        } else {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int e = 0; e < 4 + (KW > 1 ? 2 : 0); ++e) v[ci][kh][e] = 0.f;
        }
    }
    const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t4;
    const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t4) : 0;
    for (int co = 0; co < p.Cout; ++co) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const float wv = p.wp[((int64_t)(kh * KW + kw) * p.Cin_pad + ci) * p.Cout_pad + co];       // uniform: scalar load (rows ci >= Cin are zero)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += wv * v[ci][kh][e + kw];
                }
        const float os = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + co] : 1.f;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res.p) r = *reinterpret_cast<const float4*>(p.res.p + rbase + (int64_t)co * p.res.sC);
        float4 o;
        o.x = p.alpha * (p.res_scale * r.x + acc[0] * os); o.y = p.alpha * (p.res_scale * r.y + acc[1] * os);
        o.z = p.alpha * (p.res_scale * r.z + acc[2] * os); o.w = p.alpha * (p.res_scale * r.w + acc[3] * os);
        *reinterpret_cast<float4*>(p.y.p + ybase + (int64_t)co * p.y.sC) = o;
    }
}

template <typename K>
static int launch_cs(K kern, const aid_conv2d_params* p, hipStream_t st) {
    CsDev a;
    a.p = *p;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->F;
    a.tiles = aid_cdiv(p->T / 4, lpr);
    const int rpb = 256 / lpr;
    hipLaunchKernelGGL(kern, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv_small_kernel");
    return AID_OK;
}

// returns 1 if a few-channel kernel took the launch, 0 if not eligible, <0 on error
int aid_conv_small_try(const aid_conv2d_params* p, hipStream_t st) {
    if (p->act != 0 || p->epi != 0 || (p->T % 4)) return 0;
    // one thread walks the whole "other" channel dimension: ahead of the MFMA kernels on the wide, shallow levels (C <= 96, many
    // positions: 1.3-1.8x), behind them where C >= 128 and a level has too few positions to hide the serial walk
    if (p->Cin > 96 || p->Cout > 96) return 0;
    const bool k11 = p->KH == 1 && p->KW == 1, k53 = p->KH == 5 && p->KW == 3;
    if (!k11 && !k53) return 0;
    auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
    if (!al4(p->x) || !al4(p->y) || (p->res.p && !al4(p->res))) return 0;
    int rc;
    if (p->Cout <= 8 && p->Cin >= 8) {
        if (k11)            rc = p->Cout <= 2 ? launch_cs(conv_small_cout_kernel<1, 1, 2>, p, st) : launch_cs(conv_small_cout_kernel<1, 1, 8>, p, st);
        else if (p->Cout <= 2) rc = launch_cs(conv_small_cout_kernel<5, 3, 2>, p, st);
        else return 0;
    } else if (p->Cin <= 8 && p->Cout >= 8) {
        if (k11)            rc = p->Cin <= 2 ? launch_cs(conv_small_cin_kernel<1, 1, 2>, p, st) : launch_cs(conv_small_cin_kernel<1, 1, 8>, p, st);
        else if (p->Cin <= 2) rc = launch_cs(conv_small_cin_kernel<5, 3, 2>, p, st);
        else return 0;
    } else return 0;
    return rc == AID_OK ? 1 : rc;
}
