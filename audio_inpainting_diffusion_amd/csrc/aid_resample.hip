// aid_resample: 2:1 cubic FIR resampling along T (reflect padding) and the exact adjoints of both maps.
// HBM-bound; each thread produces 4 consecutive outputs (one float4 store).  Interior threads fetch their input window with
// 3-4 vector loads and evaluate the taps from registers (same terms, same order as the scalar formulas below, which remain
// the path for the threads whose window touches a row border: reflect padding / folded adjoint terms).
#include "aid_common.h"

__constant__ float c_h[8] = {-0.01171875f, -0.03515625f, 0.11328125f, 0.43359375f,
                             0.43359375f, 0.11328125f, -0.03515625f, -0.01171875f};

struct ResDev {
    aid_resample_params p;
    int Tout, lpr_log2, nrows, tiles;   // tiles = thread-tiles per row
    int xvec;                           // input view is float4-addressable: interior fast path allowed
};

__device__ __forceinline__ int refl(int i, int T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

// ---- forward maps ------------------------------------------------------------------------------------
__device__ __forceinline__ float down_at(const float* x, int T, int j) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += c_h[k] * x[refl(2 * j + k - 3, T)];
    return s;
}
__device__ __forceinline__ float up_at(const float* x, int T, int n) {
    const int m = n >> 1;
    if ((n & 1) == 0)
        return c_h[7] * x[refl(m - 2, T)] + c_h[5] * x[refl(m - 1, T)] + c_h[3] * x[refl(m, T)] + c_h[1] * x[refl(m + 1, T)];
    return c_h[6] * x[refl(m - 1, T)] + c_h[4] * x[refl(m, T)] + c_h[2] * x[refl(m + 1, T)] + c_h[0] * x[refl(m + 2, T)];
}
// ---- adjoints: gather over the (up to 3) padded-domain pre-images of input index i --------------------
// down: y[j] = sum_k h[k] xp[2j+k-3]           T = length of x (the down input), g has T/2 entries
__device__ __forceinline__ float down_adj_virtual(const float* g, int T, int ip) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int num = ip + 3 - k;
        if ((num & 1) == 0) {
            const int j = num >> 1;   // arithmetic shift: negative stays negative
            if (j >= 0 && j < (T >> 1)) s += c_h[k] * g[j];
        }
    }
    return s;
}
__device__ __forceinline__ float down_adj_at(const float* g, int T, int i) {
    float s = down_adj_virtual(g, T, i);
    if (i >= 1 && i <= 3) s += down_adj_virtual(g, T, -i);
    if (i >= T - 4 && i <= T - 2) s += down_adj_virtual(g, T, 2 * (T - 1) - i);
    return s;
}
// up: y[2m] = sum_q h[7-2q] xp[m-2+q], y[2m+1] = sum_q h[6-2q] xp[m-1+q]     T = length of x, g has 2T entries
__device__ __forceinline__ float up_adj_virtual(const float* g, int T, int ip) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m = ip + 2 - q;
        if (m >= 0 && m < T) s += c_h[7 - 2 * q] * g[2 * m];
        m = ip + 1 - q;
        if (m >= 0 && m < T) s += c_h[6 - 2 * q] * g[2 * m + 1];
    }
    return s;
}
__device__ __forceinline__ float up_adj_at(const float* g, int T, int i) {
    float s = up_adj_virtual(g, T, i);
    if (i >= 1 && i <= 2) s += up_adj_virtual(g, T, -i);
    if (i >= T - 3 && i <= T - 2) s += up_adj_virtual(g, T, 2 * (T - 1) - i);
    return s;
}

// MODE: 0 down, 1 up, 2 down-adjoint (in T/2 -> out T), 3 up-adjoint (in 2T -> out T)
template <int MODE>
__global__ __launch_bounds__(256) void resample_kernel(const ResDev a) {
    const aid_resample_params& p = a.p;
    const int tid = threadIdx.x;
    const int lpr = 1 << a.lpr_log2;
    const int sub = tid >> a.lpr_log2, lq = tid & (lpr - 1);
    const int rpb = 256 >> a.lpr_log2;
    const int tile = blockIdx.x % a.tiles;
    const int row = (blockIdx.x / a.tiles) * rpb + sub;
    if (row >= a.nrows) return;
    const int o4 = (tile * lpr + lq) * 4;
    if (o4 >= a.Tout) return;
    const int f = row % p.F;
    const int bc = row / p.F;
    const int c = bc % p.C;
    const int b = bc / p.C;
    const float* x = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)f * p.x.sF;
    float* y = p.y.p + (int64_t)b * p.y.sB + (int64_t)c * p.y.sC + (int64_t)f * p.y.sF;
    float v[4];
    if (a.xvec) {
        // ---- input window in registers, for EVERY thread (round 4).  Rounds 1-3 sent the threads whose window touches a row border down a scalar
        // path -- 2 + 2 of the T/4 threads of a row, i.e. half of them at T = 32: the adjoints of the deep levels ran at 0.85 ... 1.2 TB/s
        // (profiles/r03_pointwise_bandwidth.txt).  Now a 16-byte (8-byte) piece of the window that lies wholly inside the row is one vector load,
        // a piece outside it is filled by reflection (forward maps) or zeros (adjoints: the out-of-range terms of the transposed FIR vanish), and
        // the adjoints add the folded terms of the reflect padding for the few outputs next to a border (1 .. 3 from either end).
        const int Lin = p.T;                                 // length of the tensor passed as x
        if (MODE == 0 || MODE == 3) {
            // MODE 0: x[2*o4-4 .. 2*o4+11], output e = sum_k h[k] xp[2(o4+e)+k-3]                -> w[2e+k+1]
            // MODE 3: g[2*o4-4 .. 2*o4+11], output e = sum_q h[7-2q] g[2(o4+e+2-q)] + h[6-2q] g[2(o4+e+1-q)+1]
            //                                                                                    -> w[2e+8-2q], w[2e+7-2q]
            float w[16];
            const int base = 2 * o4 - 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i0 = base + 4 * q;
                if (i0 >= 0 && i0 + 3 < Lin) {
                    const float4 t = *reinterpret_cast<const float4*>(x + i0);
                    w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ii = i0 + j;
                        w[4 * q + j] = (MODE == 0) ? x[refl(ii, Lin)] : ((ii >= 0 && ii < Lin) ? x[ii] : 0.f);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += c_h[k] * w[2 * e + k + 1];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s += c_h[7 - 2 * q] * w[2 * e + 8 - 2 * q];
                        s += c_h[6 - 2 * q] * w[2 * e + 7 - 2 * q];
                    }
                    const int i = o4 + e;
                    if (i >= 1 && i <= 2) s += up_adj_virtual(x, a.Tout, -i);
                    if (i >= a.Tout - 3 && i <= a.Tout - 2) s += up_adj_virtual(x, a.Tout, 2 * (a.Tout - 1) - i);
                }
                v[e] = s;
            }
        } else {
            // MODE 1: x[o4/2-2 .. o4/2+3];  MODE 2: g[o4/2-2 .. o4/2+3]
            float w[6];
            const int base = (o4 >> 1) - 2;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i0 = base + 2 * q;
                if (i0 >= 0 && i0 + 1 < Lin) {
                    const float2 t = *reinterpret_cast<const float2*>(x + i0);
                    w[2 * q] = t.x; w[2 * q + 1] = t.y;
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ii = i0 + j;
                        w[2 * q + j] = (MODE == 1) ? x[refl(ii, Lin)] : ((ii >= 0 && ii < Lin) ? x[ii] : 0.f);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s;
                if (MODE == 1) {
                    const int m = e >> 1;                    // w index of x[m0 + m - 2]
                    if ((e & 1) == 0) s = c_h[7] * w[m] + c_h[5] * w[m + 1] + c_h[3] * w[m + 2] + c_h[1] * w[m + 3];
                    else              s = c_h[6] * w[m + 1] + c_h[4] * w[m + 2] + c_h[2] * w[m + 3] + c_h[0] * w[m + 4];
                } else {
                    s = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (((e + 3 - k) & 1) == 0) s += c_h[k] * w[((e + 3 - k) >> 1) + 2];     // g[(o4+e+3-k)/2]
                    const int i = o4 + e;
                    if (i >= 1 && i <= 3) s += down_adj_virtual(x, a.Tout, -i);
                    if (i >= a.Tout - 4 && i <= a.Tout - 2) s += down_adj_virtual(x, a.Tout, 2 * (a.Tout - 1) - i);
                }
                v[e] = s;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = o4 + e;
            if (MODE == 0) v[e] = down_at(x, p.T, o);
            else if (MODE == 1) v[e] = up_at(x, p.T, o);
            else if (MODE == 2) v[e] = down_adj_at(x, a.Tout, o);   // x holds g (T/2 entries), Tout = T of the down input
            else v[e] = up_adj_at(x, a.Tout, o);                     // x holds g (2T entries), Tout = T of the up input
        }
    }
    float4 r = make_float4(v[0], v[1], v[2], v[3]);
    if (p.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(y + o4);
        r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    *reinterpret_cast<float4*>(y + o4) = r;
}

extern "C" int aid_resample(const aid_resample_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->x.p && p->y.p, "aid_resample: null pointer");
    ResDev a;
    a.p = *p;
    int mode;
    if (!p->adjoint) { mode = p->up ? 1 : 0; a.Tout = p->up ? 2 * p->T : p->T / 2; }
    else             { mode = p->up ? 3 : 2; a.Tout = p->up ? p->T / 2 : 2 * p->T; }   // T = length of the incoming gradient
    AID_REQUIRE(a.Tout >= 4 && (a.Tout % 4) == 0, "aid_resample: output length must be a multiple of 4");
    AID_REQUIRE((p->y.sB % 4) == 0 && (p->y.sC % 4) == 0 && (p->y.sF % 4) == 0 && (((uintptr_t)p->y.p) & 15) == 0,
                "aid_resample: output view must be float4-addressable");
    a.xvec = ((p->x.sB % 4) == 0 && (p->x.sC % 4) == 0 && (p->x.sF % 4) == 0 && (((uintptr_t)p->x.p) & 15) == 0) ? 1 : 0;
    const int min_in = (mode == 0) ? 4 : (mode == 1 ? 3 : 0);
    AID_REQUIRE(p->T >= min_in, "aid_resample: input too short for reflect padding");
    int lpr = aid_pow2ceil(a.Tout / 4);
    if (lpr > 256) lpr = 256;
    a.lpr_log2 = aid_ilog2(lpr);
    a.nrows = p->B * p->C * p->F;
    a.tiles = aid_cdiv(a.Tout / 4, lpr);
    const int rpb = 256 / lpr;
    dim3 grid((unsigned)(aid_cdiv(a.nrows, rpb) * a.tiles));
    switch (mode) {
        case 0: hipLaunchKernelGGL(resample_kernel<0>, grid, dim3(256), 0, st, a); break;
        case 1: hipLaunchKernelGGL(resample_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(resample_kernel<2>, grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(resample_kernel<3>, grid, dim3(256), 0, st, a); break;
    }
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// =====================================================================================================
// aid_resample_poly: rational-ratio polyphase sinc resampling of whole waveforms (the tester's pre-processing,
// utils/training_utils.py:140-212 -> torchaudio.functional.resample(orig, new) with its default Hann-windowed sinc kernel).
//   y[b, i*new + j] = sum_k kernel[j][k] * xpad[b, i*orig + k],  xpad = x zero-padded by (width, width + orig)
// kernel [new][K = 2*width + orig] is built on the host (stft.py-style float64 table).  One thread per output sample;
// the K-tap windows of neighbouring outputs overlap almost completely, so the loads are L1 hits.
// =====================================================================================================
__global__ __launch_bounds__(256) void resample_poly_kernel(const aid_resample_poly_params p) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= p.Lout) return;
    const int64_t i = n / p.new_freq;
    const int j = (int)(n - i * p.new_freq);
    const float* k = p.kernel + (int64_t)j * p.K;
    const float* x = p.x + (int64_t)b * p.x_ld;
    const int64_t base = i * p.orig_freq - p.width;
    float acc = 0.f;
    for (int t = 0; t < p.K; ++t) {
        const int64_t q = base + t;
        if (q >= 0 && q < p.L) acc += k[t] * x[q];
    }
    p.y[(int64_t)b * p.y_ld + n] = acc;
}

extern "C" int aid_resample_poly(const aid_resample_poly_params* p, void* stream) {
    AID_REQUIRE(p && p->x && p->y && p->kernel, "aid_resample_poly: null pointer");
    AID_REQUIRE(p->orig_freq > 0 && p->new_freq > 0 && p->K == 2 * p->width + p->orig_freq && p->L > 0 && p->Lout > 0 && p->B > 0,
                "aid_resample_poly: bad geometry (K must be 2*width + orig_freq)");
    AID_REQUIRE(p->Lout <= (p->L * p->new_freq + p->orig_freq - 1) / p->orig_freq, "aid_resample_poly: Lout exceeds ceil(new*L/orig)");
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((p->Lout + 255) / 256), p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
