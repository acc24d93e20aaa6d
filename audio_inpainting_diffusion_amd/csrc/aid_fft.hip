// aid_fft_pass: mixed-radix Stockham FFT passes for the length-L transforms of the CQT (L = 8*7*11*13*23).
// One thread per radix-R butterfly: R strided complex loads (coalesced across threads: consecutive j),
// twiddles from a host fp64->fp32 table, naive R x R DFT in registers (R <= 32; 62 complex MACs per point over
// the five passes of L = 184184, i.e. ~0.1 GFLOP per transform -- the pass is HBM/latency-bound at 1.5 MB per
// sample), R strided stores.  Real input / Hermitian-extended input / real-part output / half-spectrum output
// are folded into the first and last pass so rfft / irfft need no extra copies.
#include "aid_common.h"

template <int R>
__global__ __launch_bounds__(256) void fft_pass_kernel(const aid_fft_pass_params p) {
    const int nb = p.N / R;                       // butterflies per batch item
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= nb) return;
    const float2* tw = reinterpret_cast<const float2*>(p.twiddle);
    const int N = p.N, Ns = p.Ns;
    const int k = j % Ns;
    const int tstep = N / (Ns * R);               // W_{Ns*R} = W_N^tstep
    const int Lh = N / 2 + 1;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int idx = j + r * nb;
        float2 x;
        if (p.in_mode == 0) x = reinterpret_cast<const float2*>(p.in)[(int64_t)b * N + idx];
        else if (p.in_mode == 1) x = make_float2(p.in[(int64_t)b * N + idx], 0.f);
        else {
            if (idx < Lh) {
                x = reinterpret_cast<const float2*>(p.in)[(int64_t)b * Lh + idx];
                if (idx == 0 || 2 * idx == N) x.y = 0.f;           // irfft ignores the imaginary part of DC / Nyquist
            } else { x = reinterpret_cast<const float2*>(p.in)[(int64_t)b * Lh + (N - idx)]; x.y = -x.y; }
        }
        if (Ns > 1 && r > 0) {
            float2 w = tw[(r * k) * tstep];       // (r*k) < R*Ns, so the index stays below N
            w.y *= -p.sign;
            x = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
        }
        v[r] = x;
    }
    float2 wr[R];                                 // W_R^m, m < R
#pragma unroll
    for (int m = 0; m < R; ++m) { wr[m] = tw[m * nb]; wr[m].y *= -p.sign; }
    const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float2 acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const float2 w = wr[(q * r) % R];
            acc.x += v[r].x * w.x - v[r].y * w.y;
            acc.y += v[r].x * w.y + v[r].y * w.x;
        }
        const int o = j0 + q * Ns;
        if (p.out_mode == 0) reinterpret_cast<float2*>(p.out)[(int64_t)b * N + o] = acc;
        else if (p.out_mode == 1) p.out[(int64_t)b * N + o] = acc.x * p.out_scale;
        else if (o < Lh) reinterpret_cast<float2*>(p.out)[(int64_t)b * Lh + o] = acc;
    }
}

template <int R>
static int launch_fft(const aid_fft_pass_params* p, hipStream_t st) {
    hipLaunchKernelGGL(fft_pass_kernel<R>, dim3((unsigned)aid_cdiv(p->N / R, 256), (unsigned)p->B), dim3(256), 0, st, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

extern "C" int aid_fft_pass(const aid_fft_pass_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->in && p->out && p->twiddle, "aid_fft_pass: null pointer");
    AID_REQUIRE(p->N > 0 && p->R >= 2 && p->N % p->R == 0 && p->Ns >= 1 && (p->N / p->R) % p->Ns == 0, "aid_fft_pass: bad factorisation");
    AID_REQUIRE(p->in != p->out, "aid_fft_pass: passes are out of place");
    switch (p->R) {
        case 2: return launch_fft<2>(p, st);   case 3: return launch_fft<3>(p, st);   case 4: return launch_fft<4>(p, st);
        case 5: return launch_fft<5>(p, st);   case 7: return launch_fft<7>(p, st);   case 8: return launch_fft<8>(p, st);
        case 11: return launch_fft<11>(p, st); case 13: return launch_fft<13>(p, st); case 16: return launch_fft<16>(p, st);
        case 17: return launch_fft<17>(p, st); case 19: return launch_fft<19>(p, st); case 23: return launch_fft<23>(p, st);
        case 29: return launch_fft<29>(p, st); case 31: return launch_fft<31>(p, st);
    }
    aid_set_error("aid_fft_pass: unsupported radix (prime factors up to 31)");
    return AID_E_BADARG;
}
