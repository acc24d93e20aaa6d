// aid_time_attention: softmax(q k^T / sqrt(F)) v along the time axis for one (sample, head) per workgroup
// column.  Sequence T <= 128, head dim F = 320..512 (frequency rows).  q/k/v are all stored [F][T] with T
// contiguous, so S = Q^T K contracts over rows and O[f][n] = sum_m V[f][m] P[n][m] keeps that layout.
//
// Round-1 version: fp32 VALU with LDS-staged F-chunks (this op is 0.03 % of the evaluation's FLOPs);
// grid = (B*H, T/32 query tiles) so that B=8 already gives 256 workgroups.
#include "aid_common.h"

#define ATT_NQ 32     // queries per workgroup
#define ATT_FC 32     // F rows staged per chunk
#define ATT_TMAX 128

__global__ __launch_bounds__(256) void time_attention_kernel(const aid_attention_params p) {
    const int bh = blockIdx.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int n0 = blockIdx.y * ATT_NQ;
    const int T = p.T, F = p.F;
    const int TP = T + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* S = sm;                      // [ATT_NQ][TP]
    float* Ks = S + ATT_NQ * TP;        // [ATT_FC][TP]   (K chunk, later V chunk)
    float* Qs = Ks + ATT_FC * TP;       // [ATT_FC][ATT_NQ]
    const int tid = threadIdx.x;
    const float* Q = p.qk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    const float* K = Q + (int64_t)F * T;
    const float* V = p.v + ((int64_t)(b * p.H + h) * F) * T;
    float* O = p.out + ((int64_t)(b * p.H + h) * F) * T;

    // ---- phase 1: S[n][m] = sum_f Q[f][n0+n] K[f][m] --------------------------------------------------
    const int nl = tid >> 3;            // 0..31 query within tile
    const int mg = tid & 7;             // key group: m = mg + 8*i
    float acc[ATT_TMAX / 8];
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) acc[i] = 0.f;
    for (int f0 = 0; f0 < F; f0 += ATT_FC) {
        for (int e = tid; e < ATT_FC * T; e += 256) {
            const int fr = e / T, m = e - fr * T;
            Ks[fr * TP + m] = (f0 + fr < F) ? K[(int64_t)(f0 + fr) * T + m] : 0.f;
        }
        for (int e = tid; e < ATT_FC * ATT_NQ; e += 256) {
            const int fr = e / ATT_NQ, n = e - fr * ATT_NQ;
            Qs[e] = (f0 + fr < F && n0 + n < T) ? Q[(int64_t)(f0 + fr) * T + n0 + n] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int fr = 0; fr < ATT_FC; ++fr) {
            const float q = Qs[fr * ATT_NQ + nl];
#pragma unroll
            for (int i = 0; i < ATT_TMAX / 8; ++i) {
                const int m = mg + 8 * i;
                if (m < T) acc[i] += q * Ks[fr * TP + m];
            }
        }
        __syncthreads();
    }
    // ---- softmax over m for each query row (8 consecutive lanes share a row) -----------------------------
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) {
        acc[i] *= p.scale;
        if (mg + 8 * i < T) mx = fmaxf(mx, acc[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) {
        if (mg + 8 * i < T) { acc[i] = expf(acc[i] - mx); sum += acc[i]; }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) {
        const int m = mg + 8 * i;
        if (m < T) {
            const float pr = acc[i] * inv;
            S[nl * TP + m] = pr;
            if (p.probs && n0 + nl < T) p.probs[(((int64_t)(b * p.H + h)) * T + n0 + nl) * T + m] = pr;
        }
    }
    __syncthreads();
    // ---- phase 2: O[f][n0+n] = sum_m V[f][m] P[n][m] ------------------------------------------------------
    const int fl = tid >> 3;            // 0..31 row within chunk
    const int ng = tid & 7;             // queries n = ng + 8*i, i < 4
    for (int f0 = 0; f0 < F; f0 += ATT_FC) {
        for (int e = tid; e < ATT_FC * T; e += 256) {
            const int fr = e / T, m = e - fr * T;
            Ks[fr * TP + m] = (f0 + fr < F) ? V[(int64_t)(f0 + fr) * T + m] : 0.f;
        }
        __syncthreads();
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < T; ++m) {
            const float v = Ks[fl * TP + m];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += v * S[(ng + 8 * i) * TP + m];
        }
        if (f0 + fl < F) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + ng + 8 * i;
                if (n < T) O[(int64_t)(f0 + fl) * T + n] = o[i];
            }
        }
        __syncthreads();
    }
}

extern "C" int aid_time_attention(const aid_attention_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->qk && p->v && p->out, "aid_time_attention: null pointer");
    AID_REQUIRE(p->T >= 1 && p->T <= ATT_TMAX, "aid_time_attention: T must be in [1,128]");
    const int TP = p->T + 1;
    const size_t lds = sizeof(float) * ((size_t)ATT_NQ * TP + (size_t)ATT_FC * TP + (size_t)ATT_FC * ATT_NQ);
    dim3 grid((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->T, ATT_NQ));
    hipLaunchKernelGGL(time_attention_kernel, grid, dim3(256), lds, st, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// =====================================================================================================
// Backward of the attention core (input-VJP of the guidance branch), T <= 128, in two kernels so that every
// CU has work (the single-kernel version ran B*H = 64 workgroups):
//   attn_bwd_ds : grid (B*H, T/32 query tiles)   dP[n][m] = sum_f dO[f][n] V[f][m]  (same tiling as the forward
//                 score phase), dS = P (dP - rowsum(P dP)) * scale -> global [B,H,T,T] (written over `probs`' twin)
//   attn_bwd_fr : grid (B*H, F/32 row chunks)    P and dS resident in LDS (2 x 66 KB at T=128); per 32 rows of F:
//                 dV[f][m] = sum_n dO[f][n] P[n][m];  dQ[f][n] = sum_m dS[n][m] K[f][m];  dK[f][m] = sum_n dS[n][m] Q[f][n]
// =====================================================================================================
__global__ __launch_bounds__(256) void attn_bwd_ds_kernel(const aid_attention_bwd_params p, float* dS) {
    const int bh = blockIdx.x;
    const int n0 = blockIdx.y * ATT_NQ;
    const int T = p.T, F = p.F, TP = T + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;                      // [ATT_FC][TP]   V chunk
    float* Qs = Ks + ATT_FC * TP;        // [ATT_FC][ATT_NQ]  dO chunk (this tile's queries)
    const int tid = threadIdx.x;
    const float* V = p.v + (int64_t)bh * F * T;
    const float* dO = p.gout + (int64_t)bh * F * T;
    const int nl = tid >> 3, mg = tid & 7;
    float acc[ATT_TMAX / 8];
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) acc[i] = 0.f;
    for (int f0 = 0; f0 < F; f0 += ATT_FC) {
        for (int e = tid; e < ATT_FC * T; e += 256) {
            const int fr = e / T, m = e - fr * T;
            Ks[fr * TP + m] = (f0 + fr < F) ? V[(int64_t)(f0 + fr) * T + m] : 0.f;
        }
        for (int e = tid; e < ATT_FC * ATT_NQ; e += 256) {
            const int fr = e / ATT_NQ, n = e - fr * ATT_NQ;
            Qs[e] = (f0 + fr < F && n0 + n < T) ? dO[(int64_t)(f0 + fr) * T + n0 + n] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int fr = 0; fr < ATT_FC; ++fr) {
            const float q = Qs[fr * ATT_NQ + nl];
#pragma unroll
            for (int i = 0; i < ATT_TMAX / 8; ++i) {
                const int m = mg + 8 * i;
                if (m < T) acc[i] += q * Ks[fr * TP + m];
            }
        }
        __syncthreads();
    }
    const int n = n0 + nl;
    if (n >= T) return;                                   // (8 consecutive lanes share a row: uniform within the group)
    const float* Prow = p.probs + ((int64_t)bh * T + n) * T;
    float pr[ATT_TMAX / 8];
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) {
        const int m = mg + 8 * i;
        pr[i] = (m < T) ? Prow[m] : 0.f;
        d += pr[i] * acc[i];
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    float* o = dS + ((int64_t)bh * T + n) * T;
#pragma unroll
    for (int i = 0; i < ATT_TMAX / 8; ++i) {
        const int m = mg + 8 * i;
        if (m < T) o[m] = pr[i] * (acc[i] - d) * p.scale;
    }
}

#define ATB_FC 32
__global__ __launch_bounds__(256) void attn_bwd_fr_kernel(const aid_attention_bwd_params p, const float* dS) {
    const int bh = blockIdx.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int f0 = blockIdx.y * ATB_FC;
    const int T = p.T, F = p.F, TP = T + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* P = sm;                 // [T][TP]
    float* G = P + T * TP;         // [T][TP]  dS
    float* A = G + T * TP;         // [ATB_FC][TP]  row chunk of dO / K / Q
    const int tid = threadIdx.x;
    const float* Q = p.qk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    const float* K = Q + (int64_t)F * T;
    const float* dO = p.gout + (int64_t)bh * F * T;
    float* dQ = p.gqk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    float* dK = dQ + (int64_t)F * T;
    float* dV = p.gv + (int64_t)bh * F * T;
    const float* Pg = p.probs + (int64_t)bh * T * T;
    const float* Sg = dS + (int64_t)bh * T * T;
    for (int e = tid; e < T * T; e += 256) { const int n = e / T, m = e - n * T; P[n * TP + m] = Pg[e]; G[n * TP + m] = Sg[e]; }
    // thread tile: 2 rows of F (fl, fl+16) x 8 columns (cg + 16*j)
    const int fl = tid >> 4, cg = tid & 15;
    auto stage = [&](const float* src) {
        __syncthreads();
        for (int e = tid; e < ATB_FC * T; e += 256) {
            const int fr = e / T, m = e - fr * T;
            A[fr * TP + m] = (f0 + fr < F) ? src[(int64_t)(f0 + fr) * T + m] : 0.f;
        }
        __syncthreads();
    };
    float o0[8], o1[8];
    // ---- dV[f][m] = sum_n dO[f][n] P[n][m] -------------------------------------------------------------------
    stage(dO);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = 0.f; o1[j] = 0.f; }
    for (int n = 0; n < T; ++n) {
        const float a0 = A[fl * TP + n], a1 = A[(fl + 16) * TP + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg + 16 * j;
            if (c < T) { const float pv = P[n * TP + c]; o0[j] += a0 * pv; o1[j] += a1 * pv; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg + 16 * j;
        if (c < T) {
            if (f0 + fl < F) { const int64_t o = (int64_t)(f0 + fl) * T + c; dV[o] = (p.accumulate_gv ? dV[o] : 0.f) + o0[j]; }
            if (f0 + fl + 16 < F) { const int64_t o = (int64_t)(f0 + fl + 16) * T + c; dV[o] = (p.accumulate_gv ? dV[o] : 0.f) + o1[j]; }
        }
    }
    // ---- dQ[f][n] = sum_m dS[n][m] K[f][m] ---------------------------------------------------------------------
    stage(K);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = 0.f; o1[j] = 0.f; }
    for (int m = 0; m < T; ++m) {
        const float a0 = A[fl * TP + m], a1 = A[(fl + 16) * TP + m];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg + 16 * j;
            if (c < T) { const float gv = G[c * TP + m]; o0[j] += a0 * gv; o1[j] += a1 * gv; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg + 16 * j;
        if (c < T) {
            if (f0 + fl < F) dQ[(int64_t)(f0 + fl) * T + c] = o0[j];
            if (f0 + fl + 16 < F) dQ[(int64_t)(f0 + fl + 16) * T + c] = o1[j];
        }
    }
    // ---- dK[f][m] = sum_n dS[n][m] Q[f][n] -----------------------------------------------------------------------
    stage(Q);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = 0.f; o1[j] = 0.f; }
    for (int n = 0; n < T; ++n) {
        const float a0 = A[fl * TP + n], a1 = A[(fl + 16) * TP + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg + 16 * j;
            if (c < T) { const float gv = G[n * TP + c]; o0[j] += a0 * gv; o1[j] += a1 * gv; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg + 16 * j;
        if (c < T) {
            if (f0 + fl < F) dK[(int64_t)(f0 + fl) * T + c] = o0[j];
            if (f0 + fl + 16 < F) dK[(int64_t)(f0 + fl + 16) * T + c] = o1[j];
        }
    }
}

extern "C" int aid_time_attention_bwd(const aid_attention_bwd_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->qk && p->v && p->probs && p->gout && p->gqk && p->gv && p->ws, "aid_time_attention_bwd: null pointer");
    AID_REQUIRE(p->T >= 1 && p->T <= ATT_TMAX, "aid_time_attention_bwd: T must be in [1,128]");
    const int TP = p->T + 1;
    const size_t lds1 = sizeof(float) * ((size_t)ATT_FC * TP + (size_t)ATT_FC * ATT_NQ);
    hipLaunchKernelGGL(attn_bwd_ds_kernel, dim3((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->T, ATT_NQ)), dim3(256), lds1, st, *p, p->ws);
    AID_CHECK_LAUNCH();
    const size_t lds2 = sizeof(float) * ((size_t)2 * p->T * TP + (size_t)ATB_FC * TP);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_bwd_fr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(attn_bwd_fr_kernel, dim3((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->F, ATB_FC)), dim3(256), lds2, st, *p, (const float*)p->ws);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
