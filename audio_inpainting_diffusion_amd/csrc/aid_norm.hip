// aid_group_stats: per-(sample, channel-group) mean / unbiased std, folded with gamma and the adaLN
// modulation into one per-(b,c) scale.  HBM-bound single read pass (float4 loads, fp64 accumulation so the
// E[x^2]-E[x]^2 form is safe), deterministic two-stage reduction (no atomics).
#include "aid_common.h"
#include "aid_wino8.h"

struct StatsDev {
    aid_group_stats_params p;
    int cg;          // channels per group
    int nrows;       // cg * F rows of T floats per group
    int lpr_log2;    // lanes per row (power of two)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void group_stats_partial(const StatsDev a) {
    const aid_group_stats_params& p = a.p;
    const int bg = blockIdx.x;              // b * groups + g
    const int split = blockIdx.y;
    const int b = bg / p.groups, g = bg - b * p.groups;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpp = 256 >> a.lpr_log2;      // rows per pass
    const int r_begin = (int)(((int64_t)a.nrows * split) / AID_STATS_SPLIT);
    const int r_end = (int)(((int64_t)a.nrows * (split + 1)) / AID_STATS_SPLIT);
    const int tq = p.T >> 2;
    double s = 0.0, ss = 0.0;
    for (int r = r_begin + sub; r < r_end; r += rpp) {
        const int c = g * a.cg + r / p.F;
        const int f = r % p.F;
        const float* row = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
        for (int q = lq; q < tq; q += lpr) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    }
    __shared__ double red[2][4];
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        double* o = p.ws + ((int64_t)bg * AID_STATS_SPLIT + split) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(256) void group_stats_final(const StatsDev a) {
    const aid_group_stats_params& p = a.p;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double n = (double)a.nrows * (double)p.T;
    const int nk = p.ws_n > 0 ? p.ws_n : AID_STATS_SPLIT;
    __shared__ double gsum[64][2];                          // (sum, sum of squares) per group (groups <= 64)
    for (int g = wave; g < p.groups; g += 4) {              // one wave per group: lane l adds partials l, l+64, ... then a fixed tree
        const double* w = p.ws + ((int64_t)(b * p.groups + g) * nk) * 2;
        double s = 0.0, ss = 0.0;
        for (int i = lane; i < nk; i += 64) { s += w[2 * i]; ss += w[2 * i + 1]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
        if (lane == 0) { gsum[g][0] = s; gsum[g][1] = ss; }
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / a.cg;
        const double s = gsum[g][0], ss = gsum[g][1];
        const double mean = s / n;
        double var = (ss - n * mean * mean) / (n - 1.0);
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / (sqrt(var) + (double)p.eps);
        const double m = p.mod ? (1.0 + (double)p.mod[(int64_t)b * p.mod_ld + c]) : 1.0;
        p.scale[(int64_t)b * p.C + c] = (float)((double)p.gamma[c] * m * inv);
        if (p.stats && c == g * a.cg) {
            p.stats[((int64_t)b * p.groups + g) * 2 + 0] = (float)mean;
            p.stats[((int64_t)b * p.groups + g) * 2 + 1] = (float)inv;
        }
    }
}
extern "C" int aid_group_stats(const aid_group_stats_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->x.p && p->gamma && p->scale && p->ws, "aid_group_stats: null pointer");
    AID_REQUIRE(p->groups > 0 && p->groups <= 64 && p->C % p->groups == 0, "aid_group_stats: C must be divisible by groups (<= 64)");
    AID_REQUIRE(p->ws_n >= 0, "aid_group_stats: bad ws_n");
    AID_REQUIRE((p->T % 4) == 0 && (p->x.sB % 4) == 0 && (p->x.sC % 4) == 0 && (p->x.sF % 4) == 0 &&
                    (((uintptr_t)p->x.p) & 15) == 0,
                "aid_group_stats: view must be float4-addressable");
    StatsDev a;
    a.p = *p;
    a.cg = p->C / p->groups;
    a.nrows = a.cg * p->F;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    if (p->ws_n == 0) {
        hipLaunchKernelGGL(group_stats_partial, dim3(p->B * p->groups, AID_STATS_SPLIT), dim3(256), 0, st, a);
        AID_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(group_stats_final, dim3(p->B), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// =====================================================================================================
// Input-VJP side: <u,v> per (sample, group) and the normalisation backward.
// =====================================================================================================
struct DotDev { aid_group_dot_params p; int cg, nrows, lpr_log2; };

__global__ __launch_bounds__(256) void group_dot_partial(const DotDev a) {
    const aid_group_dot_params& p = a.p;
    const int bg = blockIdx.x, split = blockIdx.y;
    const int b = bg / p.groups, g = bg - b * p.groups;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpp = 256 >> a.lpr_log2;
    const int r_begin = (int)(((int64_t)a.nrows * split) / AID_STATS_SPLIT);
    const int r_end = (int)(((int64_t)a.nrows * (split + 1)) / AID_STATS_SPLIT);
    const int tq = p.T >> 2;
    double s = 0.0;
    for (int r = r_begin + sub; r < r_end; r += rpp) {
        const int c = g * a.cg + r / p.F;
        const int f = r % p.F;
        const float* ru = p.u.p + (int64_t)b * p.u.sB + (int64_t)c * p.u.sC + (int64_t)f * p.u.sF;
        const float* rv = p.v.p + (int64_t)b * p.v.sB + (int64_t)c * p.v.sC + (int64_t)f * p.v.sF;
        for (int q = lq; q < tq; q += lpr) {
            const float4 x = *reinterpret_cast<const float4*>(ru + 4 * q);
            const float4 y = *reinterpret_cast<const float4*>(rv + 4 * q);
            s += (double)x.x * y.x + (double)x.y * y.y + (double)x.z * y.z + (double)x.w * y.w;
        }
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) p.ws[(int64_t)bg * AID_STATS_SPLIT + split] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int aid_group_dot(const aid_group_dot_params* p, void* stream) {
    AID_REQUIRE(p && p->u.p && p->v.p && p->ws, "aid_group_dot: null pointer");
    AID_REQUIRE(p->groups > 0 && p->C % p->groups == 0 && (p->T % 4) == 0, "aid_group_dot: bad shape");
    DotDev a;
    a.p = *p;
    a.cg = p->C / p->groups;
    a.nrows = a.cg * p->F;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    hipLaunchKernelGGL(group_dot_partial, dim3(p->B * p->groups, AID_STATS_SPLIT), dim3(256), 0, (hipStream_t)stream, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

struct NbDev { aid_norm_bwd_params p; int cg, lpr_log2, nrows, tiles; float* coef; };

// coef[b,g] = <gd,x>_g * inv / ((n-1) * std)   (written into the tail of the ws buffer as floats)
// one wave per (b, g): lane l adds partials l, l+64, ... then a fixed xor tree (deterministic)
__global__ __launch_bounds__(64) void norm_bwd_coef(const NbDev a) {
    const aid_norm_bwd_params& p = a.p;
    const int i = blockIdx.x;
    const int nk = p.ws_n > 0 ? p.ws_n : AID_STATS_SPLIT;
    double d = 0.0;
    for (int k = threadIdx.x; k < nk; k += 64) d += p.ws[(int64_t)i * nk + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
    if (threadIdx.x != 0) return;
    const double inv = (double)p.stats[2 * i + 1];
    const double sd = 1.0 / inv - (double)p.eps;
    const double n = (double)a.cg * p.F * p.T;
    a.coef[i] = (sd > 0.0) ? (float)(d * inv / ((n - 1.0) * sd)) : 0.f;
}

__global__ __launch_bounds__(256) void norm_bwd_kernel(const NbDev a) {
    const aid_norm_bwd_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    if (row >= a.nrows) return;
    const int o4 = (tile * lpr + lq) * 4;
    if (o4 >= p.T) return;
    const int f = row % p.F;
    const int bc = row / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const int bg = b * p.groups + c / a.cg;
    const float coef = a.coef[bg];
    const float mean = p.stats[2 * bg];
    const float4 gd = *reinterpret_cast<const float4*>(p.gd.p + (int64_t)b * p.gd.sB + (int64_t)c * p.gd.sC + (int64_t)f * p.gd.sF + o4);
    const float4 x = *reinterpret_cast<const float4*>(p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF + o4);
    float4 r = make_float4(gd.x - coef * (x.x - mean), gd.y - coef * (x.y - mean), gd.z - coef * (x.z - mean), gd.w - coef * (x.w - mean));
    if (p.gy.p) {
        const float4 gy = *reinterpret_cast<const float4*>(p.gy.p + (int64_t)b * p.gy.sB + (int64_t)c * p.gy.sC + (int64_t)f * p.gy.sF + o4);
        r.x += p.a * gy.x; r.y += p.a * gy.y; r.z += p.a * gy.z; r.w += p.a * gy.w;
    }
    float* o = p.out.p + (int64_t)b * p.out.sB + (int64_t)c * p.out.sC + (int64_t)f * p.out.sF + o4;
    if (p.accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(o);
        r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
    }
    *reinterpret_cast<float4*>(o) = r;
}

// Same pass with the Winograd-domain copy for the dgrad conv of the layer below.  Same thread <-> float4 mapping as norm_bwd_kernel (coalesced
// 1 KB per wave instruction); the two neighbour samples of a group of four come from the adjacent lanes by wave shuffles (recomputed from scalar
// loads at the two ends of a wave), and the group's F(4,3) input transform, times wscale[b,c], goes to `wout` (six coalesced plane stores).
__global__ __launch_bounds__(256) void norm_bwd_wino_kernel(const NbDev a) {
    const aid_norm_bwd_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    const int o4 = (tile * lpr + lq) * 4;
    const bool live = row < a.nrows && o4 < p.T;           // (all lanes stay for the shuffles)
    const int rw = live ? row : 0;
    const int f = rw % p.F;
    const int bc = rw / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const int bg = b * p.groups + c / a.cg;
    const float coef = a.coef[bg];
    const float mean = p.stats[2 * bg];
    const float* gdr = p.gd.p + (int64_t)b * p.gd.sB + (int64_t)c * p.gd.sC + (int64_t)f * p.gd.sF;
    const float* xr = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
    const float* gyr = p.gy.p ? p.gy.p + (int64_t)b * p.gy.sB + (int64_t)c * p.gy.sC + (int64_t)f * p.gy.sF : nullptr;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float4 g = *reinterpret_cast<const float4*>(gdr + o4);
        const float4 x = *reinterpret_cast<const float4*>(xr + o4);
        r = make_float4(g.x - coef * (x.x - mean), g.y - coef * (x.y - mean), g.z - coef * (x.z - mean), g.w - coef * (x.w - mean));
        if (gyr) {
            const float4 y = *reinterpret_cast<const float4*>(gyr + o4);
            r.x += p.a * y.x; r.y += p.a * y.y; r.z += p.a * y.z; r.w += p.a * y.w;
        }
        *reinterpret_cast<float4*>(p.out.p + (int64_t)b * p.out.sB + (int64_t)c * p.out.sC + (int64_t)f * p.out.sF + o4) = r;
    }
    // neighbours: sample o4-1 (= .w of the previous group) and o4+4 (= .x of the next group) of the SAME row
    const int lane = tid & 63;
    float e0 = __shfl_up(r.w, 1, 64), e5 = __shfl_down(r.x, 1, 64);
    auto one = [&](int t) {
        float v = gdr[t] - coef * (xr[t] - mean);
        if (gyr) v += p.a * gyr[t];
        return v;
    };
    if (live) {
        if (o4 == 0) e0 = 0.f;                             // the conv's zero padding
        else if (lane == 0 || lq == 0) e0 = one(o4 - 1);   // previous group lives in another wave / block
        if (o4 + 4 >= p.T) e5 = 0.f;
        else if (lane == 63 || lq == lpr - 1) e5 = one(o4 + 4);
        const float sc = p.wscale ? p.wscale[(int64_t)b * p.wscale_ld + c] : 1.f;
        e0 *= sc; e5 *= sc;
        const float e1 = r.x * sc, e2 = r.y * sc, e3 = r.z * sc, e4 = r.w * sc;
        const float s12 = e1 + e2, m12 = e1 - e2, m42 = e4 - e2, m31 = e3 - e1;
        const int G = p.T >> 2;
        float* yr = p.wout.p + (int64_t)b * p.wout.sB + (int64_t)c * p.wout.sC + (int64_t)f * p.wout.sF + (o4 >> 2);
        yr[0] = 4.f * e0 - 5.f * e2 + e4;
        yr[G] = (e3 + e4) - 4.f * s12;
        yr[2 * G] = (e4 - e3) + 4.f * m12;
        yr[3 * G] = m42 + 2.f * m31;
        yr[4 * G] = m42 - 2.f * m31;
        yr[5 * G] = 4.f * e1 - 5.f * e3 + e5;
    }
}

// F(8,3) form of the same pass (wform = 2): a thread owns one group of EIGHT samples (two float4 per operand), the neighbour samples 8g-1 and 8g+8
// come from the adjacent lanes, and the group's ten transform values go to the planes of `wout` ([B, C, F, 10, T/8]).  lpr = threads per row = T/8.
// Cooperative store of a block's Winograd-domain output staged in LDS as [row of the block][plane][lpr groups]: float4 pieces, each (row, plane)
// run of lpr floats contiguous in memory at  view(b, c, f) + plane * G + tile * lpr.   rows are (b, c, f) flattened; lpr % 4 == 0.
template <int NP>
__device__ __forceinline__ void aid_store_planes(const float* sV, const aid_view& w, int C, int F, int G, int nrows, int row0, int tile, int lpr, int rpb, int tid) {
    const int per_row = NP * lpr;
    const int n4 = rpb * per_row / 4;
    for (int i = tid; i < n4; i += 256) {
        const int e = 4 * i;
        const int rl = e / per_row, rem = e - rl * per_row;
        const int xi = rem / lpr, gq = rem - xi * lpr;
        const int row = row0 + rl;
        if (row >= nrows || tile * lpr + gq >= G) continue;
        const int f = row % F, bc = row / F;
        const int c = bc % C, b = bc / C;
        float* dst = w.p + (int64_t)b * w.sB + (int64_t)c * w.sC + (int64_t)f * w.sF + (int64_t)xi * G + tile * lpr + gq;
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(sV + e);
    }
}

// F(8,3) form of the pass (wform = 2).  Same thread <-> float4 mapping as norm_bwd_wino_kernel -- every load / store of gd, x, gy, out is one dense
// 1 KB per wave instruction -- and a PAIR of lanes (2k, 2k+1) owns one group of eight samples: the lanes swap their four values, take the two
// neighbour samples 8g-1 / 8g+8 from the lanes next to the pair, both evaluate the group's input transform (50 FMAs: cheaper than splitting it),
// and the even lane stores planes 0..4, the odd lane planes 5..9 of `wout` ([B, C, F, 10, T/8]).  (A first version gave a thread eight samples:
// its float4 accesses then stride 32 bytes across the wave, twice the cache lines per instruction -- 3.7-4.8 TB/s against 5.2-5.8 for this
// mapping, profiles/r04_pointwise_wino8.txt; staging the planes through LDS for contiguous stores changed nothing.)
__global__ __launch_bounds__(256) void norm_bwd_wino8_kernel(const NbDev a) {
    const aid_norm_bwd_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;                       // threads per row (4 samples each), >= 2
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    const int o4 = (tile * lpr + lq) * 4;
    const bool live = row < a.nrows && o4 < p.T;           // (all lanes stay for the shuffles)
    const int rw = live ? row : 0;
    const int f = rw % p.F;
    const int bc = rw / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const int bg = b * p.groups + c / a.cg;
    const float coef = a.coef[bg];
    const float mean = p.stats[2 * bg];
    const float* gdr = p.gd.p + (int64_t)b * p.gd.sB + (int64_t)c * p.gd.sC + (int64_t)f * p.gd.sF;
    const float* xr = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
    const float* gyr = p.gy.p ? p.gy.p + (int64_t)b * p.gy.sB + (int64_t)c * p.gy.sC + (int64_t)f * p.gy.sF : nullptr;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float4 g = *reinterpret_cast<const float4*>(gdr + o4);
        const float4 x = *reinterpret_cast<const float4*>(xr + o4);
        r = make_float4(g.x - coef * (x.x - mean), g.y - coef * (x.y - mean), g.z - coef * (x.z - mean), g.w - coef * (x.w - mean));
        if (gyr) {
            const float4 y = *reinterpret_cast<const float4*>(gyr + o4);
            r.x += p.a * y.x; r.y += p.a * y.y; r.z += p.a * y.z; r.w += p.a * y.w;
        }
        *reinterpret_cast<float4*>(p.out.p + (int64_t)b * p.out.sB + (int64_t)c * p.out.sC + (int64_t)f * p.out.sF + o4) = r;
    }
    const int lane = tid & 63;
    const bool odd = lane & 1;                             // (lq and lane have the same parity: lpr is even and rows start at even lanes)
    const float4 q = make_float4(__shfl_xor(r.x, 1, 64), __shfl_xor(r.y, 1, 64), __shfl_xor(r.z, 1, 64), __shfl_xor(r.w, 1, 64));     // the partner's four samples
    const float w1 = __shfl_up(r.w, 1, 64), w2 = __shfl_up(r.w, 2, 64);
    const float x1 = __shfl_down(r.x, 1, 64), x2 = __shfl_down(r.x, 2, 64);
    if (!live) return;
    const int o8 = o4 & ~7;                                // first sample of this pair's group
    float e0 = odd ? w2 : w1;                              // sample o8 - 1
    float e9 = odd ? x1 : x2;                              // sample o8 + 8
    auto one = [&](int t) {
        float v = gdr[t] - coef * (xr[t] - mean);
        if (gyr) v += p.a * gyr[t];
        return v;
    };
    const int lane0 = lane & ~1, lq0 = lq & ~1;            // the pair's even lane
    if (o8 == 0) e0 = 0.f;                                 // the conv's zero padding
    else if (lane0 == 0 || lq0 == 0) e0 = one(o8 - 1);     // previous group lives in another wave / block
    if (o8 + 8 >= p.T) e9 = 0.f;
    else if (lane0 == 62 || lq0 == lpr - 2) e9 = one(o8 + 8);
    const float sc = p.wscale ? p.wscale[(int64_t)b * p.wscale_ld + c] : 1.f;
    float d[10], V[10];
    const float4 lo = odd ? q : r, hi = odd ? r : q;       // samples o8 .. o8+3 | o8+4 .. o8+7
    d[0] = e0 * sc; d[9] = e9 * sc;
    d[1] = lo.x * sc; d[2] = lo.y * sc; d[3] = lo.z * sc; d[4] = lo.w * sc;
    d[5] = hi.x * sc; d[6] = hi.y * sc; d[7] = hi.z * sc; d[8] = hi.w * sc;
    aid_wino8_input(d, V);
    const int G = p.T >> 3;
    float* yr = p.wout.p + (int64_t)b * p.wout.sB + (int64_t)c * p.wout.sC + (int64_t)f * p.wout.sF + (o8 >> 3);
    if (!odd) {
#pragma unroll
        for (int xi = 0; xi < 5; ++xi) yr[(int64_t)xi * G] = V[xi];
    } else {
#pragma unroll
        for (int xi = 5; xi < 10; ++xi) yr[(int64_t)xi * G] = V[xi];
    }
}

int aid_w2d_input_nb(const aid_norm_bwd_params* p, const float* coef, hipStream_t st);      // aid_wino2d.hip
extern "C" int aid_norm_bwd(const aid_norm_bwd_params* p, void* stream) {
    AID_REQUIRE(p && p->gd.p && p->x.p && p->out.p && p->stats && p->ws, "aid_norm_bwd: null pointer");
    AID_REQUIRE(p->groups > 0 && p->C % p->groups == 0 && (p->T % 4) == 0, "aid_norm_bwd: bad shape");
    AID_REQUIRE(p->coef_ready == 0 || (p->coef_ready == 1 && p->ws_n > 0 && p->groups == 8), "aid_norm_bwd: coef_ready goes with conv-epilogue partials (ws_n > 0, 8 groups)");
    NbDev a;
    a.p = *p;
    a.cg = p->C / p->groups;
    // coefficients live right behind the dot partials in the caller's scratch (ws holds B*groups*SPLIT*2 doubles)
    a.coef = reinterpret_cast<float*>(const_cast<double*>(p->ws) + (int64_t)p->B * p->groups * (p->ws_n > 0 ? p->ws_n : AID_STATS_SPLIT));
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->C * p->F;
    a.tiles = aid_cdiv(p->T / 4, lpr);
    const int rpb = 256 / lpr;
    if (!p->coef_ready) {                                 // (coef_ready: written by the conv that produced the partials, aid_conv2d fin_mode = 2)
        hipLaunchKernelGGL(norm_bwd_coef, dim3(p->B * p->groups), dim3(64), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
    }
    if (p->wout.p && (p->wform == 3 || p->wform == 4))     // the 2-D forms: this pass IS the input pass of the layer below (aid_wino2d.hip), V [48 | 80][C][N]
        return aid_w2d_input_nb(p, a.coef, (hipStream_t)stream);
    if (p->wout.p && p->wform == 2) {
        AID_REQUIRE((p->T % 16) == 0 && !p->accumulate, "aid_norm_bwd: wout needs T % 16 == 0 and accumulate = 0 (neighbour samples are recomputed)");
        AID_REQUIRE(p->wout.sF >= 10 * (p->T / 8) && (p->wout.sB % 4) == 0 && (p->wout.sC % 4) == 0 && (p->wout.sF % 4) == 0 && (((uintptr_t)p->wout.p) & 15) == 0,
                    "aid_norm_bwd: wout rows (wform = 2) are [10][T/8], 16-byte aligned");
        AID_REQUIRE((p->T % 8) == 0, "aid_norm_bwd: wform = 2 needs T % 8 == 0");
        hipLaunchKernelGGL(norm_bwd_wino8_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);   // (thread <-> float4, as the F(4,3) form)
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    if (p->wout.p) {
        AID_REQUIRE(p->wform == 0 || p->wform == 1, "aid_norm_bwd: wform is 1 (F(4,3)) or 2 (F(8,3))");
        AID_REQUIRE((p->T % 16) == 0 && !p->accumulate, "aid_norm_bwd: wout needs T % 16 == 0 and accumulate = 0 (neighbour samples are recomputed)");
        AID_REQUIRE(p->wout.sF >= 6 * (p->T / 4) && (p->wout.sB % 4) == 0 && (p->wout.sC % 4) == 0 && (p->wout.sF % 4) == 0 && (((uintptr_t)p->wout.p) & 15) == 0,
                    "aid_norm_bwd: wout rows are [6][T/4], 16-byte aligned");
        hipLaunchKernelGGL(norm_bwd_wino_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    hipLaunchKernelGGL(norm_bwd_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// =====================================================================================================
// h = act(x * scale[b,c])  (one pass, float4, HBM-bound)
// =====================================================================================================
struct SaDev { aid_scale_act_params p; int lpr_log2, nrows, tiles; };

__global__ __launch_bounds__(256) void scale_act_kernel(const SaDev a) {
    const aid_scale_act_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    if (row >= a.nrows) return;
    const int o4 = (tile * lpr + lq) * 4;
    if (o4 >= p.T) return;
    const int f = row % p.F;
    const int bc = row / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const float sc = (p.scale ? p.scale[(int64_t)b * p.scale_ld + c] : 1.f) * (p.mul != 0.f ? p.mul : 1.f);
    float4 v = *reinterpret_cast<const float4*>(p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF + o4);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    if (p.act == 1) { v.x = aid_gelu(v.x); v.y = aid_gelu(v.y); v.z = aid_gelu(v.z); v.w = aid_gelu(v.w); }
    *reinterpret_cast<float4*>(p.y.p + (int64_t)b * p.y.sB + (int64_t)c * p.y.sC + (int64_t)f * p.y.sF + o4) = v;
}

// Same pass, Winograd-domain output (see aid_kernels.h): one thread produces 4 consecutive groups (16 samples) of one
// row: 4 float4 loads + the two neighbours, 6 float4 stores (one per plane xi; a row of y is [6][T/4]).
__global__ __launch_bounds__(256) void scale_act_wino_kernel(const SaDev a) {
    const aid_scale_act_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;                     // threads per row segment (16 samples each)
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    if (row >= a.nrows) return;
    const int o16 = (tile * lpr + lq) * 16;
    if (o16 >= p.T) return;
    const int f = row % p.F;
    const int bc = row / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const float sc = (p.scale ? p.scale[(int64_t)b * p.scale_ld + c] : 1.f) * (p.mul != 0.f ? p.mul : 1.f);
    const float* xr = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
    float h[18];                                         // h[0] = sample o16-1 ... h[17] = sample o16+16
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(xr + o16 + 4 * q);
        h[1 + 4 * q] = v.x; h[2 + 4 * q] = v.y; h[3 + 4 * q] = v.z; h[4 + 4 * q] = v.w;
    }
    h[0] = (o16 > 0) ? xr[o16 - 1] : 0.f;
    h[17] = (o16 + 16 < p.T) ? xr[o16 + 16] : 0.f;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        float v = h[i] * sc;
        if (p.act == 1) v = aid_gelu(v);
        h[i] = v;                                        // (gelu(0) = 0: the zero padding stays zero)
    }
    float V[6][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float e0 = h[4 * g], e1 = h[4 * g + 1], e2 = h[4 * g + 2], e3 = h[4 * g + 3], e4 = h[4 * g + 4], e5 = h[4 * g + 5];
        const float s12 = e1 + e2, m12 = e1 - e2, m42 = e4 - e2, m31 = e3 - e1;
        V[0][g] = 4.f * e0 - 5.f * e2 + e4;
        V[1][g] = (e3 + e4) - 4.f * s12;
        V[2][g] = (e4 - e3) + 4.f * m12;
        V[3][g] = m42 + 2.f * m31;
        V[4][g] = m42 - 2.f * m31;
        V[5][g] = 4.f * e1 - 5.f * e3 + e5;
    }
    const int G = p.T >> 2;
    float* yr = p.y.p + (int64_t)b * p.y.sB + (int64_t)c * p.y.sC + (int64_t)f * p.y.sF + (o16 >> 2);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
        *reinterpret_cast<float4*>(yr + (int64_t)xi * G) = make_float4(V[xi][0], V[xi][1], V[xi][2], V[xi][3]);
}

// wino = 2: the F(8,3) input transform (aid_wino8.h), y rows [10][T/8].  One thread produces 2 consecutive groups (16 samples) of one row: 4 float4
// loads + the two neighbours, 10 float2 stores (one per plane).
__global__ __launch_bounds__(256) void scale_act_wino8_kernel(const SaDev a) {
    __shared__ __attribute__((aligned(16))) float sV[256 * 20];
    const aid_scale_act_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;                     // threads per row segment (16 samples = 2 groups each)
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    const int o16 = (tile * lpr + lq) * 16;
    // T <= 64: a plane of a row is 16 / 32 bytes, and lane-by-lane stores would touch such pieces of eight rows per instruction: there the block's planes
    // are staged through LDS and written as contiguous float4 runs (4.2 / 4.5 -> 4.8 TB/s); longer rows store directly (5.6-5.9 TB/s; staged: 4.7-4.9)
    const bool staged = lpr >= 2 && lpr <= 4;
    if (row < a.nrows && o16 < p.T) {
        const int f = row % p.F;
        const int bc = row / p.F;
        const int c = bc % p.C;
        const int b = bc / p.C;
        const float sc = (p.scale ? p.scale[(int64_t)b * p.scale_ld + c] : 1.f) * (p.mul != 0.f ? p.mul : 1.f);
        const float* xr = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
        float h[18];                                     // h[0] = sample o16-1 ... h[17] = sample o16+16
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(xr + o16 + 4 * q);
            h[1 + 4 * q] = v.x; h[2 + 4 * q] = v.y; h[3 + 4 * q] = v.z; h[4 + 4 * q] = v.w;
        }
        h[0] = (o16 > 0) ? xr[o16 - 1] : 0.f;
        h[17] = (o16 + 16 < p.T) ? xr[o16 + 16] : 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            float v = h[i] * sc;
            if (p.act == 1) v = aid_gelu(v);
            h[i] = v;                                    // (gelu(0) = 0: the zero padding stays zero)
        }
        float V0[10], V1[10];
        aid_wino8_input(h, V0);
        aid_wino8_input(h + 8, V1);
        if (staged) {
#pragma unroll
            for (int xi = 0; xi < 10; ++xi)
                *reinterpret_cast<float2*>(sV + (sub * 10 + xi) * (2 * lpr) + 2 * lq) = make_float2(V0[xi], V1[xi]);
        } else {
            const int G = p.T >> 3;
            float* yr = p.y.p + (int64_t)b * p.y.sB + (int64_t)c * p.y.sC + (int64_t)f * p.y.sF + (o16 >> 3);
#pragma unroll
            for (int xi = 0; xi < 10; ++xi)
                *reinterpret_cast<float2*>(yr + (int64_t)xi * G) = make_float2(V0[xi], V1[xi]);
        }
    }
    if (!staged) return;
    __syncthreads();
    aid_store_planes<10>(sV, p.y, p.C, p.F, p.T >> 3, a.nrows, (int)(blockIdx.x / a.tiles) * rpb, tile, 2 * lpr, rpb, tid);
}

int aid_w2d_input(const aid_scale_act_params* p, hipStream_t st);      // aid_wino2d.hip

extern "C" int aid_scale_act(const aid_scale_act_params* p, void* stream) {
    AID_REQUIRE(p && p->x.p && p->y.p, "aid_scale_act: null pointer");
    AID_REQUIRE((p->T % 4) == 0, "aid_scale_act: T must be a multiple of 4");
    if (p->wino == 3 || p->wino == 4) return aid_w2d_input(p, (hipStream_t)stream);       // 2-D form F(4,5) x F(4,3): V [48][C][N]; wino = 4: F(4,5) x F(8,3), V [80][C][N]
    SaDev a;
    a.p = *p;
    if (p->wino == 2) {
        AID_REQUIRE((p->T % 16) == 0, "aid_scale_act: the Winograd-domain output needs T % 16 == 0");
        AID_REQUIRE((p->y.sB % 2) == 0 && (p->y.sC % 2) == 0 && (p->y.sF % 2) == 0 && (((uintptr_t)p->y.p) & 7) == 0 && p->y.sF >= 10 * (p->T / 8),
                    "aid_scale_act: F(8,3) Winograd-domain output rows are [10][T/8], 8-byte aligned");
        AID_REQUIRE(p->T < 32 || p->T > 64 || ((p->y.sB % 4) == 0 && (p->y.sC % 4) == 0 && (p->y.sF % 4) == 0 && (((uintptr_t)p->y.p) & 15) == 0 && (p->T % 32) == 0),
                    "aid_scale_act: F(8,3) output with T = 32 / 64 needs 16-byte aligned rows (staged float4 stores)");
        int lpr = aid_pow2ceil(p->T / 16);
        if (lpr > 256) lpr = 256;
        a.lpr_log2 = aid_ilog2(lpr);
        a.nrows = p->B * p->C * p->F;
        a.tiles = aid_cdiv(p->T / 16, lpr);
        const int rpb = 256 / lpr;
        hipLaunchKernelGGL(scale_act_wino8_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    if (p->wino) {
        AID_REQUIRE(p->wino == 1, "aid_scale_act: wino is 0, 1 (F(4,3)), 2 (F(8,3)) or 3 (F(4,5) x F(4,3))");
        AID_REQUIRE((p->T % 16) == 0, "aid_scale_act: the Winograd-domain output needs T % 16 == 0");
        AID_REQUIRE((p->y.sB % 4) == 0 && (p->y.sC % 4) == 0 && (p->y.sF % 4) == 0 && (((uintptr_t)p->y.p) & 15) == 0 && p->y.sF >= 6 * (p->T / 4),
                    "aid_scale_act: Winograd-domain output rows are [6][T/4], 16-byte aligned");
        int lpr = aid_pow2ceil(p->T / 16);
        if (lpr > 256) lpr = 256;
        a.lpr_log2 = aid_ilog2(lpr);
        a.nrows = p->B * p->C * p->F;
        a.tiles = aid_cdiv(p->T / 16, lpr);
        const int rpb = 256 / lpr;
        hipLaunchKernelGGL(scale_act_wino_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->C * p->F;
    a.tiles = aid_cdiv(p->T / 4, lpr);
    const int rpb = 256 / lpr;
    hipLaunchKernelGGL(scale_act_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
