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

struct CsDev { aid_conv2d_params p; int lpr_log2, nrows, tiles, co_per_y; };   // co_per_y: output channels walked by one blockIdx.y slice (small-Cin kernel)

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
    const int co_lo = blockIdx.y * a.co_per_y, co_hi = min(p.Cout, co_lo + a.co_per_y);
    for (int co = co_lo; co < co_hi; ++co) {
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

// ---- C -> 2, 5x3, dilation 1: the input gradient of the pyramid projections (unet...py:676) -----------------------------------------
// The one-thread-per-output-row kernel above re-reads every input row five times (once per output row that uses it) and walks all Cin
// channels serially: 0.28-0.31 ms at batch 8 where the unique data is worth 25 us.  Here a lane owns 4 samples of R CONSECUTIVE output rows and
// slides over the R + 4 input rows they need (each loaded once: (R + 4) / R re-read), the four waves of a workgroup split the input channels
// (c = wave, wave + 4, ...) and their accumulators meet in LDS, each wave finishing R / 4 of the rows.
struct CsRowsDev { aid_conv2d_params p; int lpr_log2, tiles, ngroups; };

template <int R>
__global__ __launch_bounds__(256) void conv53_cout2_rows_kernel(const CsRowsDev a) {
    constexpr int KH = 5, KW = 3, NW = 4;
    static_assert(R % NW == 0, "each wave finishes R / 4 rows");
    const aid_conv2d_params& p = a.p;
    __shared__ float red[NW][R * 8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = 1 << a.lpr_log2;
    const int g = lane >> a.lpr_log2, lq = lane & (lpr - 1);
    const int tile = blockIdx.x % a.tiles;
    const int grp = (blockIdx.x / a.tiles) * (64 >> a.lpr_log2) + g;      // group of R consecutive rows of one sample
    const bool live = grp < a.ngroups;
    const int row0 = (live ? grp : 0) * R;
    const int b = row0 / p.F, f0 = row0 - b * p.F;
    const int t4 = (tile * lpr + lq) * 4;
    float acc[R][2][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][o][e] = 0.f;
    const float* xb = p.x.p + (int64_t)b * p.x.sB;
    const bool tl = t4 > 0, tr = t4 + 4 < p.T;
    if (live && t4 < p.T) {
        for (int ci = wave; ci < p.Cin; ci += NW) {
            const float* xc = xb + (int64_t)ci * p.x.sC;
            float w[KH][KW][2];                                   // uniform: scalar loads
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const float* wp = p.wp + ((int64_t)(kh * KW + kw) * p.Cin_pad + ci) * p.Cout_pad;
                    w[kh][kw][0] = wp[0]; w[kh][kw][1] = wp[1];
                }
#pragma unroll
            for (int i = 0; i < R + KH - 1; ++i) {                // input row f0 - 2 + i feeds output rows r = i - kh
                const int fi = f0 - KH / 2 + i;
                if (fi < 0 || fi >= p.F) continue;
                const float* r_ = xc + (int64_t)fi * p.x.sF;
                const float4 c = *reinterpret_cast<const float4*>(r_ + t4);
                const float v[6] = {tl ? r_[t4 - 1] : 0.f, c.x, c.y, c.z, c.w, tr ? r_[t4 + 4] : 0.f};
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const int r = i - kh;
                    if (r < 0 || r >= R) continue;
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[r][0][e] += w[kh][kw][0] * v[e + kw];
                            acc[r][1][e] += w[kh][kw][1] * v[e + kw];
                        }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][(r * 2 + o) * 4 + e][lane] = acc[r][o][e];
    __syncthreads();
    if (!live || t4 >= p.T) return;
#pragma unroll
    for (int rr = 0; rr < R / NW; ++rr) {                          // wave w finishes rows w * R/4 .. : fixed summation order over the waves
        const int r = wave * (R / NW) + rr;
        const int f = f0 + r;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (o >= p.Cout) break;
            float s4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = (r * 2 + o) * 4 + e;
                s4[e] = ((red[0][j][lane] + red[1][j][lane]) + red[2][j][lane]) + red[3][j][lane];
            }
            const float os = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + o] : 1.f;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.res.p) q = *reinterpret_cast<const float4*>(p.res.p + (int64_t)b * p.res.sB + (int64_t)o * p.res.sC + (int64_t)f * p.res.sF + t4);
            float4 y;
            y.x = p.alpha * (p.res_scale * q.x + s4[0] * os); y.y = p.alpha * (p.res_scale * q.y + s4[1] * os);
            y.z = p.alpha * (p.res_scale * q.z + s4[2] * os); y.w = p.alpha * (p.res_scale * q.w + s4[3] * os);
            *reinterpret_cast<float4*>(p.y.p + (int64_t)b * p.y.sB + (int64_t)o * p.y.sC + (int64_t)f * p.y.sF + t4) = y;
        }
    }
}

static int launch_cout2_rows(const aid_conv2d_params* p, hipStream_t st) {
    constexpr int R = 4;
    CsRowsDev a;
    a.p = *p;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 64) lpr = 64;
    a.lpr_log2 = aid_ilog2(lpr);
    a.tiles = aid_cdiv(p->T / 4, lpr);
    a.ngroups = p->B * (p->F / R);
    const int gpb = 64 / lpr;
    hipLaunchKernelGGL(conv53_cout2_rows_kernel<R>, dim3((unsigned)(aid_cdiv(a.ngroups, gpb) * a.tiles)), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv53_cout2_rows_kernel");
    return AID_OK;
}

template <typename K>
static int launch_cs(K kern, const aid_conv2d_params* p, hipStream_t st, bool split_cout = false) {
    CsDev a;
    a.p = *p;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->F;
    a.tiles = aid_cdiv(p->T / 4, lpr);
    const int rpb = 256 / lpr;
    const int gx = aid_cdiv(a.nrows, rpb) * a.tiles;
    // small-Cin kernel at small batches: a handful of workgroups would each walk all output channels (17 us for 4 MB at B = 1) -- slice Cout over
    // blockIdx.y until ~512 workgroups exist (every slice reloads the few-channel input patch: negligible)
    int ny = 1;
    if (split_cout && gx < 512) { ny = aid_cdiv(512, gx); if (ny > p->Cout / 8) ny = p->Cout / 8; if (ny < 1) ny = 1; }
    a.co_per_y = aid_cdiv(p->Cout, ny);
    ny = aid_cdiv(p->Cout, a.co_per_y);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ny), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv_small_kernel");
    return AID_OK;
}

// returns 1 if a few-channel kernel took the launch, 0 if not eligible, <0 on error
int aid_conv_small_try(const aid_conv2d_params* p, hipStream_t st) {
    if (p->act != 0 || p->epi != 0 || (p->T % 4)) return 0;
    {   // C -> 2, 5x3, dilation 1 (any C >= 8): sliding-row kernel
        auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
        int lpr = aid_pow2ceil(p->T / 4); if (lpr > 64) lpr = 64;
        const int64_t wgs1 = (int64_t)aid_cdiv(p->F / 4, 64 / lpr) * aid_cdiv(p->T / 4, lpr);      // workgroups PER SAMPLE: the choice must not depend on the
        // batch (item b of a batch has to come out bit-identical however the batch is cut).  Deep levels: a dozen workgroups per sample walking 64
        // channels per wave lose to the MFMA kernel (244 vs 150 us at B = 1, C = 256; level at B = 8)
        if (p->KH == 5 && p->KW == 3 && p->dilF == 1 && p->Cout <= 2 && p->Cin >= 8 && !p->in_scale && (p->F % 4) == 0 && !(p->Cin > 128 && wgs1 < 16) &&
            al4(p->x) && al4(p->y) && (!p->res.p || al4(p->res))) {
            const int rc = launch_cout2_rows(p, st);
            return rc == AID_OK ? 1 : rc;
        }
    }
    // one thread walks the whole "other" channel dimension: ahead of the MFMA kernels on the wide, shallow levels (C <= 96, many
    // positions: 1.3-1.8x), behind them where C >= 128 and a level has too few positions to hide the serial walk
    if (p->Cin > 96 || (p->Cout > 96 && p->Cin > 8)) return 0;      // (few input channels: any Cout -- the walk over Cout is sliced over blockIdx.y)
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
        if (k11)            rc = p->Cin <= 2 ? launch_cs(conv_small_cin_kernel<1, 1, 2>, p, st, true) : launch_cs(conv_small_cin_kernel<1, 1, 8>, p, st, true);
        else if (p->Cin <= 2) rc = launch_cs(conv_small_cin_kernel<5, 3, 2>, p, st, true);
        else return 0;
    } else return 0;
    return rc == AID_OK ? 1 : rc;
}
