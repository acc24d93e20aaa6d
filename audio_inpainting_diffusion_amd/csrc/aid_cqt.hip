// Octave-mode NSGT filterbank kernels (analysis / synthesis / deterministic overlap-add gather).
//
//   analysis : one workgroup per (sample, band): gather the band's DFT bins times the analysis window into an
//              LDS buffer of the octave's power-of-two length T (wrapping negative offsets to the end), run a
//              radix-2 Stockham inverse FFT entirely in LDS (T <= 4096 complex = 32 KB x 2 buffers), write the
//              result straight into the planar [B,2,bins,T] tensor the U-Net kernels read.
//   synthesis: mirror image: planar coefficients -> LDS -> forward FFT -> band spectrum workspace.
//   gather   : one thread per output DFT bin sums the (2-4) bands that cover it (contiguous band range
//              precomputed on the host), applies dual window * M_k, the optional EDM preconditioning
//              (cskip*X + cout*Y) and the optional DC/Nyquist projector.  No atomics -> deterministic.
// All three are HBM/latency-bound (a few MB per sample); twiddles come from a host-computed fp64->fp32 table.
#include "aid_common.h"

#define CQT_THREADS 256

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// In-LDS Stockham radix-2 FFT of length T (power of two).  sign = -1 forward, +1 inverse (unnormalised).
// Returns the buffer holding the result.
__device__ float2* lds_fft(float2* a, float2* b, int T, int Tmax, const float2* __restrict__ tw, float sign) {
    for (int Ns = 1; Ns < T; Ns <<= 1) {
        const int tstride = Tmax / (2 * Ns);
        for (int j = threadIdx.x; j < (T >> 1); j += CQT_THREADS) {
            const int k = j & (Ns - 1);
            float2 w = tw[k * tstride];
            w.y *= -sign;                       // table holds exp(-2 pi i m/Tmax): forward as is, inverse conjugated
            const float2 u = a[j];
            const float2 v = cmul(a[j + (T >> 1)], w);
            const int j0 = ((j - k) << 1) + k;
            b[j0] = make_float2(u.x + v.x, u.y + v.y);
            b[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2* t = a; a = b; b = t;
    }
    return a;
}

__global__ __launch_bounds__(CQT_THREADS) void cqt_analysis_kernel(const aid_cqt_params p) {
    const int K = p.tab.n_oct * p.tab.bins;
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int o = k / p.tab.bins, bin = k - o * p.tab.bins;
    const int T = p.T_host[o];
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* A = lds2;
    float2* Bf = lds2 + T;
    const int Lg = p.tab.Lg[k], rc = p.tab.rc[k];
    const float* g = p.tab.g + p.tab.goff[k];
    const float2* spec = reinterpret_cast<const float2*>(p.spec) + (int64_t)b * p.Lh;
    for (int i = threadIdx.x; i < T; i += CQT_THREADS) A[i] = make_float2(0.f, 0.f);
    __syncthreads();
    const int half = Lg >> 1;
    for (int i = threadIdx.x; i < Lg; i += CQT_THREADS) {
        const int j = i - half;
        const float2 s = spec[rc + j];
        const float w = g[i];
        A[(j + T) & (T - 1)] = make_float2(s.x * w, s.y * w);
    }
    __syncthreads();
    float2* R = lds_fft(A, Bf, T, p.tab.Tmax, reinterpret_cast<const float2*>(p.tab.twiddle), +1.f);
    float sc = p.unnormalized ? 1.0f : 1.0f / (float)T;
    if (p.in_scale) sc *= p.in_scale[b];
    const aid_view& v = p.oct[o];
    float* re = v.p + (int64_t)b * v.sB + (int64_t)bin * v.sF;
    float* im = re + v.sC;
    for (int i = threadIdx.x; i < T; i += CQT_THREADS) {
        re[i] = R[i].x * sc;
        im[i] = R[i].y * sc;
    }
}

__global__ __launch_bounds__(CQT_THREADS) void cqt_synthesis_kernel(const aid_cqt_params p, int64_t ws_per_b) {
    const int K = p.tab.n_oct * p.tab.bins;
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int o = k / p.tab.bins, bin = k - o * p.tab.bins;
    const int T = p.T_host[o];
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* A = lds2;
    float2* Bf = lds2 + T;
    const aid_view& v = p.oct[o];
    const float* re = v.p + (int64_t)b * v.sB + (int64_t)bin * v.sF;
    const float* im = re + v.sC;
    for (int i = threadIdx.x; i < T; i += CQT_THREADS) A[i] = make_float2(re[i], im[i]);
    __syncthreads();
    float2* R = lds_fft(A, Bf, T, p.tab.Tmax, reinterpret_cast<const float2*>(p.tab.twiddle), -1.f);
    // band workspace offset: octaves are laid out consecutively, bins*T per octave
    int64_t woff = 0;
    for (int oo = 0; oo < o; ++oo) woff += (int64_t)p.tab.bins * p.T_host[oo];
    float2* dst = reinterpret_cast<float2*>(p.band_ws) + (int64_t)b * ws_per_b + woff + (int64_t)bin * T;
    for (int i = threadIdx.x; i < T; i += CQT_THREADS) dst[i] = R[i];
}

__global__ __launch_bounds__(256) void cqt_gather_kernel(const aid_cqt_gather_params p) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (v >= p.Lh) return;
    float2 acc = make_float2(0.f, 0.f);
    if (p.band_ws) {
        const float2* ws = reinterpret_cast<const float2*>(p.band_ws) + (int64_t)b * p.ws_per_b;
        const int k0 = p.kfirst[v], kn = p.kcount[v];
        for (int i = 0; i < kn; ++i) {
            const int k = k0 + i;
            const int j = v - p.rc[k];
            const int Lg = p.Lg[k];
            const int idx = j + (Lg >> 1);
            if (idx < 0 || idx >= Lg) continue;
            const int T = p.Tk[k];
            const float2 c = ws[p.woff[k] + ((j + T) & (T - 1))];
            const float w = p.gdM[p.goff[k] + idx];
            acc.x += c.x * w;
            acc.y += c.y * w;
        }
        if (p.cout) { const float co = p.cout[b]; acc.x *= co; acc.y *= co; }
        if (p.band_scale) { const float bs = p.band_scale[v]; acc.x *= bs; acc.y *= bs; }
    }
    if (p.X) {
        const float2 x = reinterpret_cast<const float2*>(p.X)[(int64_t)b * p.Lh + v];
        const float cs = p.cskip ? p.cskip[b] : 1.f;
        acc.x += cs * x.x;
        acc.y += cs * x.y;
    }
    if (p.hpf) { const float h = p.hpf[v]; acc.x *= h; acc.y *= h; }
    reinterpret_cast<float2*>(p.Y)[(int64_t)b * p.Lh + v] = acc;
}

static int cqt_check(const aid_cqt_params* p) {
    AID_REQUIRE(p && p->tab.n_oct > 0 && p->tab.n_oct <= AID_CQT_MAX_OCT, "aid_cqt: bad octave count");
    for (int o = 0; o < p->tab.n_oct; ++o) {
        AID_REQUIRE(p->T_host[o] >= 2 && p->T_host[o] <= 8192 && (p->T_host[o] & (p->T_host[o] - 1)) == 0,
                    "aid_cqt: octave lengths must be powers of two in [2, 8192]");
        AID_REQUIRE(p->T_host[o] <= p->tab.Tmax, "aid_cqt: twiddle table too short");
        AID_REQUIRE(p->oct[o].p != nullptr, "aid_cqt: null octave view");
    }
    return AID_OK;
}

extern "C" int aid_cqt_analysis(const aid_cqt_params* p, void* stream) {
    int rc = cqt_check(p);
    if (rc) return rc;
    AID_REQUIRE(p->spec, "aid_cqt_analysis: null spectrum");
    const int K = p->tab.n_oct * p->tab.bins;
    int Tmax = 0;
    for (int o = 0; o < p->tab.n_oct; ++o) Tmax = p->T_host[o] > Tmax ? p->T_host[o] : Tmax;
    const size_t lds = sizeof(float2) * 2 * Tmax;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)cqt_analysis_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(cqt_analysis_kernel, dim3(p->B * K), dim3(CQT_THREADS), lds, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

extern "C" int aid_cqt_synthesis(const aid_cqt_params* p, void* stream) {
    int rc = cqt_check(p);
    if (rc) return rc;
    AID_REQUIRE(p->band_ws, "aid_cqt_synthesis: null workspace");
    const int K = p->tab.n_oct * p->tab.bins;
    int Tmax = 0;
    int64_t per_b = 0;
    for (int o = 0; o < p->tab.n_oct; ++o) { Tmax = p->T_host[o] > Tmax ? p->T_host[o] : Tmax; per_b += (int64_t)p->tab.bins * p->T_host[o]; }
    const size_t lds = sizeof(float2) * 2 * Tmax;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)cqt_synthesis_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(cqt_synthesis_kernel, dim3(p->B * K), dim3(CQT_THREADS), lds, (hipStream_t)stream, *p, per_b);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

extern "C" int aid_cqt_gather(const aid_cqt_gather_params* p, void* stream) {
    AID_REQUIRE(p && p->Y && (p->band_ws || p->X), "aid_cqt_gather: null pointer");
    hipLaunchKernelGGL(cqt_gather_kernel, dim3(aid_cdiv(p->Lh, 256), p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
