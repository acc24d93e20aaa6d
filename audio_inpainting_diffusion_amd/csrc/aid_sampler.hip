// Element-wise kernels of the EDM sampling loop (HBM-trivial: a few [B,L] fp32 arrays per call).
#include "aid_common.h"

__global__ __launch_bounds__(256) void axpby_kernel(const aid_axpby_params p) {
    const int b = blockIdx.y;
    const float a = p.a ? p.a[b] : 1.f, bb = p.b ? p.b[b] : 1.f;
    const int64_t base = (int64_t)b * p.L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.L; i += (int64_t)gridDim.x * 256) {
        float v = a * p.x[base + i];
        if (p.y) v += bb * p.y[base + i];
        p.out[base + i] = v;
    }
}

extern "C" int aid_axpby(const aid_axpby_params* p, void* stream) {
    AID_REQUIRE(p && p->x && p->out, "aid_axpby: null pointer");
    int gx = aid_cdiv(p->L, 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(axpby_kernel, dim3(gx, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

__global__ __launch_bounds__(256) void score_step_kernel(const aid_score_step_params p) {
    const int b = blockIdx.y;
    const float t = p.t[b], h = p.h[b];
    const float inv_t = 1.0f / t;
    const int64_t base = (int64_t)b * p.L;
    const float* sm = p.smask ? p.smask + (int64_t)b * p.smask_sB : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.L; i += (int64_t)gridDim.x * 256) {
        const float x = p.x[base + i];
        float xh = p.xhat[base + i];
        if (sm) { const float m = sm[i]; xh = m * p.yobs[base + i] + (1.f - m) * xh; }
        if (p.xh_out) p.xh_out[base + i] = xh;
        // score = (xh - x)/t^2 ; d = -t*score = (x - xh)/t      (edm_sampler_inpainting.py:105,230)
        const float d = (x - xh) * inv_t;
        if (p.mode == 0) {
            if (p.dout) p.dout[base + i] = d;
            p.xnext[base + i] = x + h * d;
        } else {
            p.xnext[base + i] = p.x0[base + i] + h * (0.5f * p.d0[base + i] + 0.5f * d);
        }
    }
}

extern "C" int aid_score_step(const aid_score_step_params* p, void* stream) {
    AID_REQUIRE(p && p->x && p->xhat && p->t && p->h && p->xnext, "aid_score_step: null pointer");
    AID_REQUIRE(p->mode == 0 || (p->x0 && p->d0), "aid_score_step: Heun combine needs x0 and d0");
    AID_REQUIRE(!p->smask || p->yobs, "aid_score_step: projection needs the observations");
    int gx = aid_cdiv(p->L, 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(score_step_kernel, dim3(gx, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ---- y = a*u + b*v on strided [B,C,F,T] views (float4 along T) ------------------------------------------
struct Add2Dev { aid_add2_params p; int lpr_log2, nrows, tiles; };

__global__ __launch_bounds__(256) void add2_kernel(const Add2Dev a) {
    const aid_add2_params& p = a.p;
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
    const float4 u = *reinterpret_cast<const float4*>(p.u.p + (int64_t)b * p.u.sB + (int64_t)c * p.u.sC + (int64_t)f * p.u.sF + o4);
    float4 r = make_float4(p.a * u.x, p.a * u.y, p.a * u.z, p.a * u.w);
    if (p.v.p) {
        const float4 v = *reinterpret_cast<const float4*>(p.v.p + (int64_t)b * p.v.sB + (int64_t)c * p.v.sC + (int64_t)f * p.v.sF + o4);
        r.x += p.b * v.x; r.y += p.b * v.y; r.z += p.b * v.z; r.w += p.b * v.w;
    }
    *reinterpret_cast<float4*>(p.y.p + (int64_t)b * p.y.sB + (int64_t)c * p.y.sC + (int64_t)f * p.y.sF + o4) = r;
}

extern "C" int aid_add2(const aid_add2_params* p, void* stream) {
    AID_REQUIRE(p && p->u.p && p->y.p, "aid_add2: null pointer");
    AID_REQUIRE((p->T % 4) == 0, "aid_add2: T must be a multiple of 4");
    Add2Dev a;
    a.p = *p;
    int lpr = aid_pow2ceil(p->T / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->C * p->F;
    a.tiles = aid_cdiv(p->T / 4, lpr);
    const int rpb = 256 / lpr;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles)), dim3(256), 0, (hipStream_t)stream, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ---- per-item L2 norms and the analytic seed of the reconstruction-guidance gradient --------------------
__device__ __forceinline__ double block_sum_1024(double v) {
    __shared__ double red[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) { for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i]; red[0] = t; }
    __syncthreads();
    t = red[0];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(1024) void row_norm_kernel(const aid_row_norm_params p) {
    const int b = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) { const double v = p.x[(int64_t)b * p.L + i]; s += v * v; }
    s = block_sum_1024(s);
    if (threadIdx.x == 0) p.out[b] = (float)sqrt(s);
}

extern "C" int aid_row_norm(const aid_row_norm_params* p, void* stream) {
    AID_REQUIRE(p && p->x && p->out, "aid_row_norm: null pointer");
    hipLaunchKernelGGL(row_norm_kernel, dim3(p->B), dim3(1024), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

__global__ __launch_bounds__(1024) void guidance_seed_kernel(const aid_guidance_seed_params p) {
    const int b = blockIdx.x;
    const float* m = p.mask + (int64_t)b * p.mask_sB;
    const int64_t base = (int64_t)b * p.L;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) {
        const double r = (double)p.y[base + i] - (double)(m[i] * p.xhat[base + i]);
        s += r * r;
    }
    s = block_sum_1024(s);
    const float nrm = (float)sqrt(s);
    if (threadIdx.x == 0) p.norm[b] = nrm;
    const float inv = nrm > 0.f ? 1.0f / nrm : 0.f;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) {
        const float mi = m[i];
        p.g[base + i] = -mi * (p.y[base + i] - mi * p.xhat[base + i]) * inv;
    }
}

extern "C" int aid_guidance_seed(const aid_guidance_seed_params* p, void* stream) {
    AID_REQUIRE(p && p->xhat && p->y && p->mask && p->g && p->norm, "aid_guidance_seed: null pointer");
    hipLaunchKernelGGL(guidance_seed_kernel, dim3(p->B), dim3(1024), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
