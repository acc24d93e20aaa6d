// aid_time_attention / aid_time_attention_bwd: softmax(q k^T / sqrt(F)) v along the time axis, one (sample, head) per
// workgroup column, on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  Sequence T <= 128, head dim F = 320..512
// frequency rows (any F, T are accepted: out-of-range operands are loaded as zeros).  q/k/v are all stored [F][T] with T
// contiguous, which is exactly what the MFMA operand layouts want:
//
//   32x32x2 MFMA:  A is 32(M) x 2(K): lane l holds A[l%32][l/32];  B is 2(K) x 32(N): lane l holds B[l/32][l%32];
//                  D[i][j] sits in lane (j, half) register r with i = (r&3) + 8*(r>>2) + 4*half          (half = l/32)
//
//   scores, TRANSPOSED   S^T[m][n] = sum_f K[f][m] Q[f][n]:  A-lane (m, f+half) and B-lane (n, f+half) are both coalesced
//     row loads of the [F][T] arrays.  With keys on M, a lane owns ONE query n and 16 keys: the softmax over keys is a
//     reduction over the lane's registers, one wavefront shuffle across the two half-waves (xor 32) and one LDS exchange
//     between the four waves (one key tile each) -- no transposes, no [T][T] round trip through memory.
//   O[f][n] = sum_m V[f][m] P[n][m]:  the K dimension (keys) of a 32x32x2 step pairs one operand column from each
//     half-wave; pairing keys (m, m+4) lets a lane fetch the four A columns of four consecutive steps with ONE 16-byte
//     load of V[f][8j + 4*half ..] and makes the B fragment the softmax registers' own layout (exchanged through LDS so
//     that every wave sees all keys).
// The VJP (guidance branch) is the same algebra: dP^T = V^T dO (scores kernel shape), dS = P (dP - rowsum(P dP)) scale with
// the same register / shuffle / LDS reductions, then dV = dO P, dQ = K dS^T, dK = Q dS as three [32 rows of F] x [T] x [T]
// MFMA products per workgroup reading P and dS straight from L2.
#include "aid_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ATT_TMAX 128
#define ATT_LDP 33            // leading dimension of the P^T exchange tile in LDS (conflict-free column access)

__device__ __forceinline__ float4 att_ld4(const float* __restrict__ row, int c, int T, bool row_ok, bool vec_ok) {
    // four consecutive columns c..c+3 of one row; zeros outside [0,T) or when the row does not exist
    if (!row_ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok && c + 3 < T) return *reinterpret_cast<const float4*>(row + c);
    float4 v;
    v.x = (c + 0 < T) ? row[c + 0] : 0.f;
    v.y = (c + 1 < T) ? row[c + 1] : 0.f;
    v.z = (c + 2 < T) ? row[c + 2] : 0.f;
    v.w = (c + 3 < T) ? row[c + 3] : 0.f;
    return v;
}

__device__ __forceinline__ void att_st4(float* __restrict__ row, int c, int T, bool vec_ok, float4 v) {
    if (vec_ok && c + 3 < T) { *reinterpret_cast<float4*>(row + c) = v; return; }
    if (c + 0 < T) row[c + 0] = v.x;
    if (c + 1 < T) row[c + 1] = v.y;
    if (c + 2 < T) row[c + 2] = v.z;
    if (c + 3 < T) row[c + 3] = v.w;
}

// acc = sum_f A[f][ma] * Bm[f][nb]  as a 32x32 tile (rows: columns ma0.. of A, cols: columns nb0.. of Bm); A, Bm are [F][T]
__device__ __forceinline__ f32x16 att_scores_tile(const float* __restrict__ A, const float* __restrict__ Bm, int F, int T,
                                                  int ma, int nb, int half) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool aok = ma < T, bok = nb < T;
    const float* ap = A + ma;
    const float* bp = Bm + nb;
    for (int f = 0; f < F; f += 16) {                        // 8 k-steps per trip: all 16 loads in flight before the first MFMA
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int fr = f + 2 * u + half;
            a[u] = (aok && fr < F) ? ap[(int64_t)fr * T] : 0.f;
            b[u] = (bok && fr < F) ? bp[(int64_t)fr * T] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    return acc;
}

// =====================================================================================================
// forward: grid (B*H, ceil(T/32) query tiles), 4 waves; wave w owns keys [32w, 32w+32)
// =====================================================================================================
__global__ __launch_bounds__(256) void time_attention_kernel(const aid_attention_params p) {
    const int bh = blockIdx.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int n0 = blockIdx.y * 32;
    const int T = p.T, F = p.F;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = ((T & 3) == 0) && ((((uintptr_t)p.v) | ((uintptr_t)p.probs)) & 15) == 0;
    __shared__ float red[2][4][32];
    __shared__ float Pt[ATT_TMAX * ATT_LDP];                 // P^T[m][n] of this query tile
    const float* Q = p.qk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    const float* K = Q + (int64_t)F * T;
    const float* V = p.v + (int64_t)bh * F * T;
    float* O = p.out + (int64_t)bh * F * T;

    const int m0 = wave * 32;
    const int n = n0 + l32;
    f32x16 acc;
    if (m0 < T) acc = att_scores_tile(K, Q, F, T, m0 + l32, n, half);     // S^T[m][n]
    else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    if (p.bias && n < T && m0 < T) {                        // sim = (q k^T + bias[h][n][m]) * scale   (use_rel_pos, unet...py:364-366)
        const float* brow = p.bias + ((int64_t)h * T + n) * T;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mb = m0 + 8 * q + 4 * half;
            const float4 bb = att_ld4(brow, mb, T, mb < T, vec && ((((uintptr_t)p.bias) & 15) == 0));
            acc[4 * q] += bb.x; acc[4 * q + 1] += bb.y; acc[4 * q + 2] += bb.z; acc[4 * q + 3] += bb.w;
        }
    }
    // ---- softmax over the keys of query n: registers -> half-waves (shuffle) -> waves (LDS) ----------------------------
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        acc[r] *= p.scale;
        if (m < T) mx = fmaxf(mx, acc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (half == 0) red[0][wave][l32] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0][l32], red[0][1][l32]), fmaxf(red[0][2][l32], red[0][3][l32]));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        acc[r] = (m < T) ? expf(acc[r] - mx) : 0.f;
        sum += acc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    if (half == 0) red[1][wave][l32] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[1][0][l32] + red[1][1][l32] + red[1][2][l32] + red[1][3][l32]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mb = m0 + 8 * q + 4 * half;
        const float4 pr = make_float4(acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv);
        Pt[(mb + 0) * ATT_LDP + l32] = pr.x;                  // (every wave writes its 32 rows: rows >= T hold zeros)
        Pt[(mb + 1) * ATT_LDP + l32] = pr.y;
        Pt[(mb + 2) * ATT_LDP + l32] = pr.z;
        Pt[(mb + 3) * ATT_LDP + l32] = pr.w;
        if (p.probs && n < T && mb < T) att_st4(p.probs + ((int64_t)bh * T + n) * T, mb, T, vec, pr);
    }
    __syncthreads();
    // ---- O[f][n] = sum_m V[f][m] P[n][m]: B fragments for all keys, pairing (m, m+4) --------------------------------------
    float bf[ATT_TMAX / 2];
#pragma unroll
    for (int j = 0; j < ATT_TMAX / 8; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[4 * j + s] = (8 * j < T) ? Pt[(8 * j + s + 4 * half) * ATT_LDP + l32] : 0.f;
    const int ntile = (F + 31) >> 5;
    for (int ft = wave; ft < ntile; ft += 4) {
        const int row = ft * 32 + l32;
        const float* vr = V + (int64_t)row * T;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int j = 0; j < ATT_TMAX / 8; ++j) {
            if (8 * j < T) {                                  // (wave-uniform)
                const float4 a = att_ld4(vr, 8 * j + 4 * half, T, row < F, vec);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[4 * j + 0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[4 * j + 1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[4 * j + 2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[4 * j + 3], o, 0, 0, 0);
            }
        }
        if (n < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < F) O[(int64_t)f * T + n] = o[r];
            }
        }
    }
}

extern "C" int aid_time_attention(const aid_attention_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->qk && p->v && p->out, "aid_time_attention: null pointer");
    AID_REQUIRE(p->T >= 1 && p->T <= ATT_TMAX, "aid_time_attention: T must be in [1,128]");
    AID_REQUIRE(p->B > 0 && p->H > 0 && p->F > 0, "aid_time_attention: empty shape");
    dim3 grid((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->T, 32));
    hipLaunchKernelGGL(time_attention_kernel, grid, dim3(256), 0, st, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// =====================================================================================================
// Backward of the attention core (input-VJP of the guidance branch), T <= 128, two kernels:
//   attn_bwd_ds : grid (B*H, T/32 query tiles)   dP^T[m][n] = sum_f V[f][m] dO[f][n]  (the forward's score product),
//                 dS = P (dP - rowsum(P dP)) * scale -> global [B,H,T,T] (caller scratch `ws`)
//   attn_bwd_fr : grid (B*H, row blocks of F)     per 32 rows of F and 32 columns of T (one wave each):
//                 dV[f][m] = sum_n dO[f][n] P[n][m];  dQ[f][n] = sum_m K[f][m] dS[n][m];  dK[f][m] = sum_n Q[f][n] dS[n][m]
//                 A fragments: one 16-byte row load per four k-steps (pairing (k, k+4)); B fragments: rows of P / dS
//                 (coalesced) or, for dQ, 16-byte loads along the rows of dS -- all straight from L2.
// =====================================================================================================
__global__ __launch_bounds__(256) void attn_bwd_ds_kernel(const aid_attention_bwd_params p, float* __restrict__ dS) {
    const int bh = blockIdx.x;
    const int n0 = blockIdx.y * 32;
    const int T = p.T, F = p.F;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = ((T & 3) == 0) && ((((uintptr_t)p.probs) | ((uintptr_t)dS)) & 15) == 0;
    __shared__ float red[4][32];
    const float* V = p.v + (int64_t)bh * F * T;
    const float* dO = p.gout + (int64_t)bh * F * T;
    const int m0 = wave * 32;
    const int n = n0 + l32;
    f32x16 acc;
    if (m0 < T) acc = att_scores_tile(V, dO, F, T, m0 + l32, n, half);    // dP^T[m][n]
    else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    const float* Prow = p.probs + ((int64_t)bh * T + n) * T;
    float pr[16];
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mb = m0 + 8 * q + 4 * half;
        const float4 v = att_ld4(Prow, mb, T, n < T && mb < T, vec);
        pr[4 * q] = v.x; pr[4 * q + 1] = v.y; pr[4 * q + 2] = v.z; pr[4 * q + 3] = v.w;
        d += (v.x * acc[4 * q] + v.y * acc[4 * q + 1]) + (v.z * acc[4 * q + 2] + v.w * acc[4 * q + 3]);
    }
    d += __shfl_xor(d, 32, 64);
    if (half == 0) red[wave][l32] = d;
    __syncthreads();
    d = (red[0][l32] + red[1][l32]) + (red[2][l32] + red[3][l32]);
    if (n >= T) return;
    float* o = dS + ((int64_t)bh * T + n) * T;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mb = m0 + 8 * q + 4 * half;
        if (mb < T)
            att_st4(o, mb, T, vec, make_float4(pr[4 * q] * (acc[4 * q] - d) * p.scale, pr[4 * q + 1] * (acc[4 * q + 1] - d) * p.scale,
                                               pr[4 * q + 2] * (acc[4 * q + 2] - d) * p.scale, pr[4 * q + 3] * (acc[4 * q + 3] - d) * p.scale));
    }
}

// NCOL = column tiles of 32 covering T (1, 2 or 4); the 4 waves split into 4/NCOL row tiles x NCOL column tiles
template <int NCOL>
__global__ __launch_bounds__(256) void attn_bwd_fr_kernel(const aid_attention_bwd_params p, const float* __restrict__ dS) {
    constexpr int RT = 4 / NCOL;                              // row tiles (32 rows of F) per workgroup
    const int bh = blockIdx.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int T = p.T, F = p.F;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % NCOL, rt = wave / NCOL;
    const int f0 = (blockIdx.y * RT + rt) * 32;
    const int c0 = ct * 32;
    if (f0 >= F || c0 >= T) return;                           // (no barriers below)
    const float* Q = p.qk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    const float* K = Q + (int64_t)F * T;
    const float* dO = p.gout + (int64_t)bh * F * T;
    float* dQ = p.gqk + ((int64_t)b * p.H * 2 * F + (int64_t)h * 2 * F) * T;
    float* dK = dQ + (int64_t)F * T;
    float* dV = p.gv + (int64_t)bh * F * T;
    const float* Pg = p.probs + (int64_t)bh * T * T;
    const float* Sg = dS + (int64_t)bh * T * T;
    const bool vec = ((T & 3) == 0) && ((((uintptr_t)p.qk) | ((uintptr_t)p.gout) | ((uintptr_t)dS)) & 15) == 0;
    const int row = f0 + l32;                                 // A-fragment row of this lane
    const bool rok = row < F;
    const int col = c0 + l32;                                 // B-fragment / output column of this lane
    const bool cok = col < T;
    f32x16 aV, aQ, aK;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aV[r] = 0.f; aQ[r] = 0.f; aK[r] = 0.f; }
    const float* dOr = dO + (int64_t)row * T;
    const float* Kr = K + (int64_t)row * T;
    const float* Qr = Q + (int64_t)row * T;
    const float* Srow = Sg + (int64_t)col * T;                // dS[n = col][.]
    for (int j = 0; 8 * j < T; ++j) {
        const int k0 = 8 * j + 4 * half;                      // this half-wave's four contraction indices k0..k0+3
        const float4 a_do = att_ld4(dOr, k0, T, rok, vec);
        const float4 a_k = att_ld4(Kr, k0, T, rok, vec);
        const float4 a_q = att_ld4(Qr, k0, T, rok, vec);
        const float4 b_s = att_ld4(Srow, k0, T, cok, vec);    // dS[col][k0..k0+3]: B of dQ (contraction over keys m = k)
        float bp[4], bs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = cok && (k0 + s) < T;
            bp[s] = ok ? Pg[(int64_t)(k0 + s) * T + col] : 0.f;      // P[n = k][m = col]
            bs[s] = ok ? Sg[(int64_t)(k0 + s) * T + col] : 0.f;      // dS[n = k][m = col]
        }
        aV = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.x, bp[0], aV, 0, 0, 0);
        aV = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.y, bp[1], aV, 0, 0, 0);
        aV = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.z, bp[2], aV, 0, 0, 0);
        aV = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.w, bp[3], aV, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_k.x, b_s.x, aQ, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_k.y, b_s.y, aQ, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_k.z, b_s.z, aQ, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_k.w, b_s.w, aQ, 0, 0, 0);
        aK = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.x, bs[0], aK, 0, 0, 0);
        aK = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.y, bs[1], aK, 0, 0, 0);
        aK = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.z, bs[2], aK, 0, 0, 0);
        aK = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.w, bs[3], aK, 0, 0, 0);
    }
    if (!cok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = f0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (f >= F) continue;
        const int64_t o = (int64_t)f * T + col;
        dV[o] = (p.accumulate_gv ? dV[o] : 0.f) + aV[r];
        dQ[o] = aQ[r];
        dK[o] = aK[r];
    }
}

extern "C" int aid_time_attention_bwd(const aid_attention_bwd_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->qk && p->v && p->probs && p->gout && p->gqk && p->gv && p->ws, "aid_time_attention_bwd: null pointer");
    AID_REQUIRE(p->T >= 1 && p->T <= ATT_TMAX, "aid_time_attention_bwd: T must be in [1,128]");
    AID_REQUIRE(p->B > 0 && p->H > 0 && p->F > 0, "aid_time_attention_bwd: empty shape");
    hipLaunchKernelGGL(attn_bwd_ds_kernel, dim3((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->T, 32)), dim3(256), 0, st, *p, p->ws);
    AID_CHECK_LAUNCH();
    const int ncol = p->T <= 32 ? 1 : (p->T <= 64 ? 2 : 4);
    const dim3 grid((unsigned)(p->B * p->H), (unsigned)aid_cdiv(p->F, 32 * (4 / ncol)));
    if (ncol == 1) hipLaunchKernelGGL(attn_bwd_fr_kernel<1>, grid, dim3(256), 0, st, *p, (const float*)p->ws);
    else if (ncol == 2) hipLaunchKernelGGL(attn_bwd_fr_kernel<2>, grid, dim3(256), 0, st, *p, (const float*)p->ws);
    else hipLaunchKernelGGL(attn_bwd_fr_kernel<4>, grid, dim3(256), 0, st, *p, (const float*)p->ws);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
