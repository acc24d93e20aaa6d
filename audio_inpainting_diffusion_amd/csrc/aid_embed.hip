// aid_embed: RFF noise embedding + 3-layer ReLU MLP (one workgroup per sample; a few kFLOP).
// aid_modulation: all affine/gate Linears of the network as one [B,E] x [E,N] product; HBM-bound on the
// stacked weight matrix (~30 MB for the 22 kHz net), one wave per output row, all B samples per wave.
#include "aid_common.h"

#define EMB_MAX 1024

__global__ __launch_bounds__(256) void embed_kernel(const aid_embed_params p) {
    __shared__ float bufA[EMB_MAX];
    __shared__ float bufB[EMB_MAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float s = p.sigma[b];
    for (int i = tid; i < p.rff; i += 256) {
        const float ang = 2.0f * 3.14159265358979323846f * s * p.rff_freq[i];   // 2*np.pi*sigma*freqs, float32 like torch
        bufA[i] = sinf(ang);
        bufA[p.rff + i] = cosf(ang);
    }
    __syncthreads();
    const int d0 = 2 * p.rff;
    for (int o = tid; o < p.h0; o += 256) {
        float acc = p.b0[o];
        for (int i = 0; i < d0; ++i) acc += bufA[i] * p.w0[(int64_t)o * d0 + i];
        bufB[o] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int o = tid; o < p.h1; o += 256) {
        float acc = p.b1[o];
        for (int i = 0; i < p.h0; ++i) acc += bufB[i] * p.w1[(int64_t)o * p.h0 + i];
        bufA[o] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int o = tid; o < p.E; o += 256) {
        float acc = p.b2[o];
        for (int i = 0; i < p.h1; ++i) acc += bufA[i] * p.w2[(int64_t)o * p.h1 + i];
        p.emb[(int64_t)b * p.E + o] = fmaxf(acc, 0.f);
    }
}

extern "C" int aid_embed(const aid_embed_params* p, void* stream) {
    AID_REQUIRE(p && p->sigma && p->emb, "aid_embed: null pointer");
    AID_REQUIRE(2 * p->rff <= EMB_MAX && p->h0 <= EMB_MAX && p->h1 <= EMB_MAX, "aid_embed: layer too wide");
    hipLaunchKernelGGL(embed_kernel, dim3(p->B), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

#define MOD_BMAX 16
__global__ __launch_bounds__(256) void modulation_kernel(const aid_modulation_params p, int b0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p.N) return;
    const int nb = min(MOD_BMAX, p.B - b0);
    float acc[MOD_BMAX];
#pragma unroll
    for (int i = 0; i < MOD_BMAX; ++i) acc[i] = 0.f;
    for (int e = lane; e < p.E; e += 64) {
        const float w = p.W[(int64_t)row * p.E + e];
#pragma unroll
        for (int i = 0; i < MOD_BMAX; ++i)
            if (i < nb) acc[i] += w * p.emb[(int64_t)(b0 + i) * p.E + e];
    }
#pragma unroll
    for (int i = 0; i < MOD_BMAX; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && i < nb) p.mod[(int64_t)(b0 + i) * p.N + row] = v + (p.bias ? p.bias[row] : 0.f);
    }
}

extern "C" int aid_modulation(const aid_modulation_params* p, void* stream) {
    AID_REQUIRE(p && p->emb && p->W && p->mod, "aid_modulation: null pointer");
    for (int b0 = 0; b0 < p->B; b0 += MOD_BMAX) {
        hipLaunchKernelGGL(modulation_kernel, dim3(aid_cdiv(p->N, 4)), dim3(256), 0, (hipStream_t)stream, *p, b0);
        AID_CHECK_LAUNCH();
    }
    return AID_OK;
}
