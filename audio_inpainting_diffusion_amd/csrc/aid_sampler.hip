// Element-wise kernels of the EDM sampling loop (HBM-trivial: a few [B,L] fp32 arrays per call).
#include "aid_common.h"

__global__ __launch_bounds__(256) void axpby_kernel(const aid_axpby_params p) {
    const int b = blockIdx.y;
    const float a = p.a ? p.a[b] : p.a_host, bb = p.b ? p.b[b] : p.b_host;
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
    const float t = p.t ? p.t[b] : p.t_host, h = p.h ? p.h[b] : p.h_host;
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
    AID_REQUIRE(p && p->x && p->xhat && p->xnext, "aid_score_step: null pointer");
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
    const float* m = p.mask ? p.mask + (int64_t)b * p.mask_sB : nullptr;
    const int64_t base = (int64_t)b * p.L;
    const int kind = p.norm_type;                        // 2: || r ||_2, 1: || r ||_1, 3: sum smooth_l1(r; beta)   with r = y - mask*xhat
    const float beta = p.beta;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) {
        const float mi = m ? m[i] : 1.f;
        const double r = (double)p.y[base + i] - (double)(mi * p.xhat[base + i]);
        if (kind == 2) s += r * r;
        else if (kind == 1) s += fabs(r);
        else { const double ar = fabs(r); s += (ar < (double)beta) ? 0.5 * r * r / (double)beta : ar - 0.5 * (double)beta; }
    }
    s = block_sum_1024(s);
    const float nrm = kind == 2 ? (float)sqrt(s) : (float)s;
    if (threadIdx.x == 0) p.norm[b] = nrm;
    const float inv = nrm > 0.f ? 1.0f / nrm : 0.f;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) {
        const float mi = m ? m[i] : 1.f;
        const float r = p.y[base + i] - mi * p.xhat[base + i];
        float d;                                         // d norm / d r
        if (kind == 2) d = r * inv;
        else if (kind == 1) d = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
        else d = (fabsf(r) < beta) ? r / beta : ((r > 0.f) ? 1.f : -1.f);
        p.g[base + i] = -mi * d;
    }
}

extern "C" int aid_guidance_seed(const aid_guidance_seed_params* p, void* stream) {
    AID_REQUIRE(p && p->xhat && p->y && p->g && p->norm, "aid_guidance_seed: null pointer");
    AID_REQUIRE(p->norm_type == 2 || p->norm_type == 1 || (p->norm_type == 3 && p->beta > 0.f), "aid_guidance_seed: norm_type is 2, 1 or 3 (smooth-L1, beta > 0)");
    hipLaunchKernelGGL(guidance_seed_kernel, dim3(p->B), dim3(1024), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ---- guidance step: out = xhat - coef / (||g||_2 * inv_sqrt_len + eps) * g, per item (edm_sampler_inpainting.py:83-97) ----
__global__ __launch_bounds__(1024) void guidance_step_kernel(const aid_guidance_step_params p) {
    const int b = blockIdx.x;
    const int64_t base = (int64_t)b * p.L;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) { const double v = p.g[base + i]; s += v * v; }
    s = block_sum_1024(s);
    const float normguide = (float)sqrt(s) * p.inv_sqrt_len;
    const float sc = p.coef / (normguide + p.eps);
    if (threadIdx.x == 0 && p.s_out) p.s_out[b] = sc;
    for (int64_t i = threadIdx.x; i < p.L; i += 1024) {
        const float u = sc * p.g[base + i];
        if (p.step_out) p.step_out[base + i] = u;
        p.out[base + i] = p.xhat[base + i] - u;
    }
}

extern "C" int aid_guidance_step(const aid_guidance_step_params* p, void* stream) {
    AID_REQUIRE(p && p->xhat && p->g && p->out, "aid_guidance_step: null pointer");
    hipLaunchKernelGGL(guidance_step_kernel, dim3(p->B), dim3(1024), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ---- host scalars -> device rows: out[i*ld + b] = v[i] for b < B (the EDM preconditioning scalars of one evaluation in ONE launch) ----
__global__ void set_rows_kernel(const aid_set_rows_params p) {
    const int i = blockIdx.x;
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) p.out[(int64_t)i * p.ld + b] = p.v[i];
}

extern "C" int aid_set_rows(const aid_set_rows_params* p, void* stream) {
    AID_REQUIRE(p && p->out && p->n >= 1 && p->n <= 8 && p->B >= 1 && p->ld >= p->B, "aid_set_rows: 1..8 rows of B <= ld values");
    hipLaunchKernelGGL(set_rows_kernel, dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
