// aid_group_stats: per-(sample, channel-group) mean / unbiased std, folded with gamma and the adaLN
// modulation into one per-(b,c) scale.  HBM-bound single read pass (float4 loads, fp64 accumulation so the
// E[x^2]-E[x]^2 form is safe), deterministic two-stage reduction (no atomics).
#include "aid_common.h"

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
    const double n = (double)a.nrows * (double)p.T;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        const int g = c / a.cg;
        const double* w = p.ws + ((int64_t)(b * p.groups + g) * AID_STATS_SPLIT) * 2;
        double s = 0.0, ss = 0.0;
        for (int i = 0; i < AID_STATS_SPLIT; ++i) { s += w[2 * i]; ss += w[2 * i + 1]; }
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
    AID_REQUIRE(p->groups > 0 && p->C % p->groups == 0, "aid_group_stats: C must be divisible by groups");
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
    hipLaunchKernelGGL(group_stats_partial, dim3(p->B * p->groups, AID_STATS_SPLIT), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    hipLaunchKernelGGL(group_stats_final, dim3(p->B), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
