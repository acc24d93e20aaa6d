// STFT-domain masking operator of the spectrogram-inpainting degradation and its exact adjoint.
//
//   A(x) = crop_L( istft( M .* stft( zero-pad(x) ) ) )       (testing/edm_sampler_inpainting.py:271-290)
//
// written as  A = D * sum_n R'_n^T W C_n W R_n :  R_n = reflect-centred frame extraction, W = window,
// C_n = F^-1 diag(m_n) F (real symmetric circulant because m_n is real and mirrored), R'_n = plain frame
// extraction, D = 1/window-envelope and crop.  The adjoint runs the SAME two kernels with the roles of the
// boundary handling swapped:  A^T = sum_n R_n^T W C_n W R'_n D.
//
//   frames : one workgroup per (sample, frame): gather (reflect or zero outside, optional 1/envelope pre-scale)
//            times window -> LDS radix-2 Stockham FFT -> times mask column (mirrored) -> inverse FFT -> times
//            window / n_fft -> frame buffer [B, n_frames, n_fft].
//   ola    : one thread per output sample gathers the (n_fft/hop) frames covering it (no atomics ->
//            deterministic); forward: times 1/envelope, optional `c0*A(x) + add1 + add2` epilogue (the
//            projection y + x - A(x) of :360); adjoint: folds the reflected borders back.
// HBM-trivial (a few [B,L] arrays and one [B, 4L] frame buffer per call).
#include "aid_common.h"

#define STFT_THREADS 256

__device__ __forceinline__ float2 st_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// radix-2 Stockham in LDS; tw = exp(-2 pi i m / N), m < N/2.  sign -1 forward, +1 inverse (unnormalised).
__device__ float2* st_fft(float2* a, float2* b, int N, const float2* __restrict__ tw, float sign) {
    for (int Ns = 1; Ns < N; Ns <<= 1) {
        const int tstride = N / (2 * Ns);
        for (int j = threadIdx.x; j < (N >> 1); j += STFT_THREADS) {
            const int k = j & (Ns - 1);
            float2 w = tw[k * tstride];
            w.y *= -sign;
            const float2 u = a[j];
            const float2 v = st_cmul(a[j + (N >> 1)], w);
            const int j0 = ((j - k) << 1) + k;
            b[j0] = make_float2(u.x + v.x, u.y + v.y);
            b[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2* t = a; a = b; b = t;
    }
    return a;
}

__global__ __launch_bounds__(STFT_THREADS) void stft_frames_kernel(const aid_stft_params p) {
    const int b = blockIdx.x / p.n_frames, n = blockIdx.x - b * p.n_frames;
    const int N = p.n_fft, half = N >> 1;
    extern __shared__ __attribute__((aligned(16))) float2 st_lds[];
    float2* A = st_lds;
    float2* Bf = st_lds + N;
    const float* x = p.x + (int64_t)b * p.L;
    const int64_t start = (int64_t)n * p.hop - half;
    for (int j = threadIdx.x; j < N; j += STFT_THREADS) {
        int64_t q = start + j;
        float v = 0.f;
        bool in = true;
        if (p.adjoint) {
            in = (q >= 0 && q < p.Lp);                       // R'_n: nothing outside the (trimmed) istft support
        } else {
            if (q < 0) q = -q;                               // R_n: torch.stft(center=True, pad_mode='reflect')
            if (q >= p.Lp) q = 2 * (p.Lp - 1) - q;
        }
        if (in && q < p.L) {
            v = x[q];
            if (p.adjoint) v *= p.inv_env[q];
        }
        A[j] = make_float2(v * p.window[j], 0.f);
    }
    __syncthreads();
    const float2* tw = reinterpret_cast<const float2*>(p.twiddle);
    float2* R = st_fft(A, Bf, N, tw, -1.f);
    const float* m = p.mask + (int64_t)b * p.mask_sB + n;
    for (int f = threadIdx.x; f < N; f += STFT_THREADS) {
        const int fm = f <= half ? f : N - f;
        const float mv = m[(int64_t)fm * p.mask_ld];
        R[f].x *= mv;
        R[f].y *= mv;
    }
    __syncthreads();
    float2* S = st_fft(R, R == A ? Bf : A, N, tw, +1.f);
    float* fr = p.frames + ((int64_t)b * p.n_frames + n) * N;
    const float inv = 1.0f / (float)N;
    for (int j = threadIdx.x; j < N; j += STFT_THREADS) fr[j] = S[j].x * p.window[j] * inv;
}

// sum of the frames covering centred position q = pos + n_fft/2
__device__ __forceinline__ float st_cover(const float* __restrict__ fr, int64_t pos, int N, int hop, int n_frames) {
    const int64_t q = pos + (N >> 1);
    int64_t n1 = q / hop;
    int64_t n0 = (q - (N - 1) + hop - 1) / hop;
    if (q - (N - 1) < 0) n0 = 0;
    if (n1 > n_frames - 1) n1 = n_frames - 1;
    float acc = 0.f;
    for (int64_t n = n0; n <= n1; ++n) acc += fr[n * N + (q - n * hop)];
    return acc;
}

__global__ __launch_bounds__(256) void stft_ola_kernel(const aid_stft_params p) {
    const int b = blockIdx.y;
    const float* fr = p.frames + (int64_t)b * p.n_frames * p.n_fft;
    const int N = p.n_fft, half = N >> 1;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < p.L; t += (int64_t)gridDim.x * 256) {
        float v = st_cover(fr, t, N, p.hop, p.n_frames);
        if (p.adjoint) {
            if (t >= 1 && t <= half) v += st_cover(fr, -t, N, p.hop, p.n_frames);              // left reflection
            const int64_t r = 2 * (p.Lp - 1) - t;
            if (r >= p.Lp && r < p.Lp + half) v += st_cover(fr, r, N, p.hop, p.n_frames);       // right reflection
        } else {
            v *= p.inv_env[t];
        }
        v *= p.c0;
        if (p.add1) v += p.add1[(int64_t)b * p.L + t];
        if (p.add2) v += p.add2[(int64_t)b * p.L + t];
        p.out[(int64_t)b * p.L + t] = v;
    }
}

static int stft_check(const aid_stft_params* p) {
    AID_REQUIRE(p && p->frames && p->window && p->mask && p->twiddle, "aid_stft: null pointer");
    AID_REQUIRE(p->n_fft >= 16 && p->n_fft <= 8192 && (p->n_fft & (p->n_fft - 1)) == 0, "aid_stft: n_fft must be a power of two in [16, 8192]");
    AID_REQUIRE(p->hop > 0 && p->hop <= p->n_fft, "aid_stft: bad hop");
    AID_REQUIRE(p->Lp >= p->L && p->Lp > p->n_fft / 2 && p->Lp % p->hop == 0, "aid_stft: padded length must be a multiple of hop, >= L and > n_fft/2");
    AID_REQUIRE(p->n_frames == 1 + p->Lp / p->hop, "aid_stft: n_frames must be 1 + Lp/hop");
    AID_REQUIRE(p->inv_env, "aid_stft: the window envelope is needed in both directions");
    return AID_OK;
}

extern "C" int aid_stft_frames(const aid_stft_params* p, void* stream) {
    int rc = stft_check(p);
    if (rc) return rc;
    AID_REQUIRE(p->x, "aid_stft_frames: null input");
    const size_t lds = (size_t)p->n_fft * 2 * sizeof(float2);
    hipLaunchKernelGGL(stft_frames_kernel, dim3(p->B * p->n_frames), dim3(STFT_THREADS), lds, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

extern "C" int aid_stft_ola(const aid_stft_params* p, void* stream) {
    int rc = stft_check(p);
    if (rc) return rc;
    AID_REQUIRE(p->out, "aid_stft_ola: null output");
    int gx = aid_cdiv(p->L, 256); if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(stft_ola_kernel, dim3(gx, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
