// Parameter-gradient kernels of the training step (diff_params/edm.py:166-193 loss_fn,
// training/trainer.py:253-304 train_step / update_ema).  The activation gradients come from the input-VJP plan that the
// guidance branch already runs; these kernels add what only training needs:
//
//   aid_conv2d_wgrad   P[b,s][co][tap][ci] = alpha * sum_{f,t} gy[b,co,f,t] * in[b,ci,f+(kh-KH/2)*dil,t+kw-KW/2]
//                      per-(sample, row-split) partial weight gradients on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//                      M = 32 output channels, N = 32 input channels, K = positions (t); one accumulator tile per tap, the
//                      taps of a layer spread over the 4 waves of a workgroup.  gy and the KH dilated input rows of a 64-sample
//                      chunk are staged in LDS (rows padded to an odd length: the transposed fragment reads are conflict-free).
//   aid_wgrad_reduce   dW[co,ci,tap] += sum_b gate[b,co] * in_scale[b,ci] * sum_s P ;
//                      dgate[b,co]   = sum_{ci,tap} W[co,ci,tap] * in_scale[b,ci] * sum_s P      (= alpha <gy, conv output>: the
//                      gate's gradient needs the un-gated conv output, which the forward never stores -- it is recovered from the
//                      per-sample partials instead of recomputing the convolution)
//   aid_channel_dot    S[b,c] = sum_{f,t} u*v  (per channel; fp64 accumulation)     -> gradient of the per-(b,c) norm/adaLN scale
//   aid_scale_bwd      S / scale -> d gamma[c] (accumulated over b), d affine[b,c] (into the modulation-gradient row)
//   aid_modulation_bwd / aid_embed_bwd   the stacked affine/gate Linears and the RFF-MLP, backwards
//   aid_adam / aid_ema / aid_sumsq       fused optimiser update, EMA update and gradient-norm partials over flat buffers
// Everything is deterministic (fixed summation orders, no atomics).
#include "aid_common.h"
#include "aid_wino8.h"
#include "aid_wino45.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WG_TC 64                 // positions (t) per staged chunk
#define WG_LDA (WG_TC + 1)       // gy rows in LDS
#define WG_LDB (WG_TC + 3)       // input rows in LDS: one halo sample each side (KW <= 3) + pad to an odd length
#define WG_MAXKH 5
#define WG_MAXTAPS 15

struct WgDev {
    aid_conv2d_wgrad_params p;
    int co_tiles, ci_tiles;
};

#define WG_NW 8                  // waves per workgroup
#define WG_QPW 2                 // taps per wave (WG_NW * WG_QPW >= 15)
#define WG_THREADS (64 * WG_NW)
#define WG_NS (WG_MAXKH + 1)     // ring slots of input rows: the KH rows of the current step + the new row of the next one

// One workgroup per CU (85 KB of LDS, eight waves = two per SIMD).  Staging is asynchronous INSIDE the workgroup: the next step's gy
// tile and new input row are loaded into registers before the K loop, written to the other gy buffer / the free ring slot half-way
// through it, and one barrier per step separates the steps.  (Two workgroups per CU with synchronous staging ran in lockstep -- equal
// step lengths never let their phases drift apart -- so staging time simply added to the MFMA time: profiles/r02_wgrad_gates.txt.)
template <int COB>     // 32-row output-channel blocks per workgroup tile: 2 (64 channels), or 3 for layers whose Cout is an odd multiple of 32 (C = 96)
__global__ __launch_bounds__(WG_THREADS, 2) void conv_wgrad_kernel(const WgDev a) {
    constexpr int WG_CO = 32 * COB;
    const aid_conv2d_wgrad_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = (blockIdx.x % a.co_tiles) * WG_CO;
    const int ci0 = (blockIdx.x / a.co_tiles) * 32;
    const int b = blockIdx.y / p.S, s = blockIdx.y - b * p.S;
    const int KH = p.KH, KW = p.KW, ntaps = KH * KW;
    const int kwc = KW / 2, khc = KH / 2;
    const int NS = KH + 1;
    const int nco = min(COB, (p.Cout - co0 + 31) / 32);   // 32-row blocks of this tile that have rows (wave-uniform)

    __shared__ float gyT0[WG_CO * WG_LDA + 8];            // (+8: the fast path reads one k-step past the chunk)
    __shared__ float gyT1[WG_CO * WG_LDA + 8];
    __shared__ float xT[WG_NS * 32 * WG_LDB + 8];

    // taps of this wave: wave, wave + 8; COB accumulator tiles (co0.., co0+32.., ...) per tap
    f32x16 acc[WG_QPW][COB];
#pragma unroll
    for (int q = 0; q < WG_QPW; ++q)
#pragma unroll
        for (int i = 0; i < COB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;
    int tkh[WG_QPW], tkw[WG_QPW];
#pragma unroll
    for (int q = 0; q < WG_QPW; ++q) { const int tap = wave + WG_NW * q; tkh[q] = tap / KW; tkw[q] = tap - tkh[q] * KW; }

    const bool vec = ((p.T & 3) == 0) && ((p.gy.sB | p.gy.sC | p.gy.sF | p.x.sB | p.x.sC | p.x.sF) & 3) == 0 &&
                     ((((uintptr_t)p.gy.p) | ((uintptr_t)p.x.p)) & 15) == 0;
    auto ld4 = [&](const float* base, bool ok, int t) {   // four consecutive samples t..t+3 of one row (zero outside)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && t < p.T) {
            const float* src = base + t;
            if (vec && t + 3 < p.T) v = *reinterpret_cast<const float4*>(src);
            else { v.x = src[0]; if (t + 1 < p.T) v.y = src[1]; if (t + 2 < p.T) v.z = src[2]; if (t + 3 < p.T) v.w = src[3]; }
        }
        return v;
    };
    // This thread's share of a step's staging: gy 32 COB rows x 16 float4 -> COB per thread; one input row 32 x 16 float4 -> 1 per thread (+ halo)
    const int q4 = tid & 15, srow = tid >> 4;             // float4 column, base row 0..31 (gy rows srow + 32 i)
    auto load_gy = [&](int f, int t0, float4 (&g)[COB]) {
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            const int co = co0 + srow + 32 * i;
            g[i] = ld4(p.gy.p + (int64_t)b * p.gy.sB + (int64_t)co * p.gy.sC + (int64_t)f * p.gy.sF, co < p.Cout, t0 + 4 * q4);
        }
    };
    auto store_gy = [&](float* gyT, const float4 (&g)[COB]) {
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            float* d = gyT + (srow + 32 * i) * WG_LDA + 4 * q4;
            d[0] = g[i].x; d[1] = g[i].y; d[2] = g[i].z; d[3] = g[i].w;
        }
    };
    auto load_row = [&](int fi, int t0, float4& x, float& hl) {
        const bool fok = fi >= 0 && fi < p.F;
        const int ci = ci0 + srow;
        x = ld4(p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF, fok && ci < p.Cin, t0 + 4 * q4);
        hl = 0.f;
        if (KW > 1 && tid < 64) {
            const int side = tid & 1, cih = ci0 + (tid >> 1);
            const int t = side ? (t0 + WG_TC) : (t0 - 1);
            if (cih < p.Cin && fok && t >= 0 && t < p.T) hl = p.x.p[(int64_t)b * p.x.sB + (int64_t)cih * p.x.sC + (int64_t)fi * p.x.sF + t];
        }
    };
    auto store_row = [&](int slot, const float4& x, float hl) {
        float* d = xT + (slot * 32 + srow) * WG_LDB + 1 + 4 * q4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        if (KW > 1 && tid < 64) xT[(slot * 32 + (tid >> 1)) * WG_LDB + ((tid & 1) ? (WG_TC + 1) : 0)] = hl;
    };

    // ---- the step sequence ------------------------------------------------------------------------------------------------------
    // One step = one output row f x one chunk of 64 positions.  Rows are walked along the dilation sub-lattice (f, f + dil, ...): consecutive
    // steps of such a chain share KH-1 of their KH input rows, which stay in a ring of LDS slots (chain row m = n + kh in slot m % NS) -- one
    // new row per step.  Order: chunk, residue class, row of the class; the S splits of a sample cut this sequence (nct * F steps) into equal
    // contiguous parts, so large dilations keep their chains (a split that starts inside a chain stages all KH rows once).
    const int dil = (KH > 1) ? p.dilF : 1;
    const int nct = (p.T + WG_TC - 1) / WG_TC;
    const int tc = (p.T < WG_TC) ? ((p.T + 3) & ~3) : WG_TC;     // positions actually walked per chunk (short levels: T = 32)
    const int qd = p.F / dil, rem = p.F - qd * dil;             // classes res < rem have qd + 1 rows, the others qd
    const int64_t total = (int64_t)nct * p.F;
    const int g_lo = (int)((total * s) / p.S), g_hi = (int)((total * (s + 1)) / p.S);
    int c = g_lo / p.F, res, n;
    {
        const int r = g_lo - c * p.F;
        if (r < rem * (qd + 1)) { res = r / (qd + 1); n = r - res * (qd + 1); }
        else { const int r2 = r - rem * (qd + 1); res = rem + r2 / qd; n = r2 - (res - rem) * qd; }
    }
    float4 pg[COB], px;
    float ph = 0.f;
    auto stage_chain_start = [&](int f, int t0, int n0) {       // all KH rows of a step (synchronous)
        for (int kh = 0; kh < KH; ++kh) {
            load_row(f + (kh - khc) * dil, t0, px, ph);
            store_row((n0 + kh) % NS, px, ph);
        }
    };
    if (g_lo < g_hi) {
        load_gy(res + n * dil, c * WG_TC, pg);
        store_gy(gyT0, pg);
        stage_chain_start(res + n * dil, c * WG_TC, n);
    }
    __syncthreads();
    int cur = 0;
    for (int g = g_lo; g < g_hi; ++g) {
        // ---- next step: its loads are issued now and land in registers while this step multiplies ---------------------------------------
        int n2 = n + 1, res2 = res, c2 = c;
        if (n2 >= qd + (res < rem ? 1 : 0)) { n2 = 0; ++res2; if (res2 >= dil || res2 >= p.F) { res2 = 0; ++c2; } }
        const bool have = g + 1 < g_hi;
        const bool chain = n2 != 0;
        const int f2 = res2 + n2 * dil;
        if (have) {
            load_gy(f2, c2 * WG_TC, pg);
            if (chain) load_row(f2 + (KH - 1 - khc) * dil, c2 * WG_TC, px, ph);
        }
        const float* gyT = cur ? gyT1 : gyT0;
        float* gyN = cur ? gyT0 : gyT1;
        auto store_next = [&]() {
            if (have) {
                store_gy(gyN, pg);
                if (chain) store_row((n2 + KH - 1) % NS, px, ph);
            }
        };
        // ---- K loop over the chunk's positions: A = gy^T (rows: co), B = shifted input (cols: ci) ------------------------------------
        int boff[WG_QPW];
#pragma unroll
        for (int q = 0; q < WG_QPW; ++q) boff[q] = (((n + tkh[q]) % NS) * 32 + l32) * WG_LDB + 1 + half + tkw[q] - kwc;
        const float* ap = gyT + l32 * WG_LDA + half;
        if (ntaps == 15 && (tc & 15) == 0) {
            // 5x3 fast path: every wave runs two taps (wave 7's second is a dummy whose tile is never written: its SIMD slot would idle
            // otherwise), fragments of k-step j+1 are read while step j multiplies (reads past tc land in the row padding / array tail)
            const float* bp0 = xT + boff[0];
            const float* bp1 = xT + boff[1];
            float av[2][COB], bv[2][2];
            auto ld = [&](int k, int u) {
#pragma unroll
                for (int i = 0; i < COB; ++i)
                    if (i < nco) av[u][i] = ap[i * 32 * WG_LDA + k];
                bv[u][0] = bp0[k]; bv[u][1] = bp1[k];
            };
            ld(0, 0);
            const int kmid = tc >> 1;
            for (int k0 = 0; k0 < tc; k0 += 8) {
                if (k0 == kmid) store_next();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ld(k0 + 2 * j + 2, (j + 1) & 1);
#pragma unroll
                    for (int q = 0; q < WG_QPW; ++q) {
#pragma unroll
                        for (int i = 0; i < COB; ++i)
                            if (i < nco) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][i], bv[j & 1][q], acc[q][i], 0, 0, 0);
                    }
                }
            }
        } else if (ntaps == 1) {
            // 1x1 layers: one tap, so the eight waves split the chunk's positions instead (k-steps wave, wave + 8, ...) and their accumulators
            // are added in wave order after the last step
            store_next();
            const int b0 = ((n % NS) * 32 + l32) * WG_LDB + 1 + half;      // tap 0 for every wave (boff[] is per-wave taps)
            for (int k = 2 * wave; k < tc; k += 2 * WG_NW) {
                const float bvv = xT[b0 + k];
#pragma unroll
                for (int i = 0; i < COB; ++i)
                    if (i < nco) acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[i * 32 * WG_LDA + k], bvv, acc[0][i], 0, 0, 0);
            }
        } else {
            const int kmid = (tc >> 2) << 1;
            for (int k = 0; k < tc; k += 2) {
                if (k == kmid) store_next();
                float av1[COB];
#pragma unroll
                for (int i = 0; i < COB; ++i) av1[i] = (i < nco) ? ap[i * 32 * WG_LDA + k] : 0.f;
#pragma unroll
                for (int q = 0; q < WG_QPW; ++q) {
                    if (wave + WG_NW * q < ntaps) {       // (wave-uniform)
                        const float bvv = xT[boff[q] + k];
#pragma unroll
                        for (int i = 0; i < COB; ++i)
                            if (i < nco) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[i], bvv, acc[q][i], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                  // this step's fragment reads and the next step's stores are done
        if (have && !chain) {                             // the next step starts a chain: its KH rows, synchronously
            stage_chain_start(f2, c2 * WG_TC, n2);
            __syncthreads();
        }
        cur ^= 1;
        n = n2; res = res2; c = c2;
    }
    if (ntaps == 1) {                                       // fixed-order sum of the eight waves' tiles through LDS (the staging arrays are free now)
        float* red = xT;                                    // [wave][16][64] floats = 32 KB per output-channel block
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[0][i][r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = red[r * 64 + lane];
                    for (int w = 1; w < WG_NW; ++w) v += red[(w * 16 + r) * 64 + lane];
                    acc[0][i][r] = v;
                }
            }
        }
    }
    // ---- write the partials P[(b*S+s)][co][tap][ci] (ci fastest: every store instruction writes two full 128-byte lines) ---------------
    float* P = p.P + ((int64_t)(b * p.S + s) * ntaps) * p.Cout * p.Cin;
    const int ci = ci0 + l32;
#pragma unroll
    for (int q = 0; q < WG_QPW; ++q) {
        const int tap = wave + WG_NW * q;
        if (tap >= ntaps || ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < COB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.Cout) P[((int64_t)co * ntaps + tap) * p.Cin + ci] = p.alpha * acc[q][i][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// 1x1 layers: dW[co,ci] = sum over positions of gy[co,pos] x[ci,pos] -- a plain GEMM with K = positions.  One tap means nothing to spread over the
// waves, so a workgroup owns a (32 COB) x (32 NCI) block of dW, every wave holds ALL its COB x NCI accumulator tiles and the eight waves split the
// positions of a chunk (k-steps wave, wave + 8, ...); a fixed-order sum over the waves follows the last step.  Both operands are double-buffered
// in LDS (transposed reads, odd pitch), the next step is prefetched into registers and stored half-way through the K loop.
template <int COB, int NCI>
__global__ __launch_bounds__(WG_THREADS, 2) void conv_wgrad_1x1_kernel(const WgDev a) {
    constexpr int NCO = 32 * COB, NCX = 32 * NCI;
    const aid_conv2d_wgrad_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = (blockIdx.x % a.co_tiles) * NCO;
    const int ci0 = (blockIdx.x / a.co_tiles) * NCX;
    const int b = blockIdx.y / p.S, s = blockIdx.y - b * p.S;
    const int nco = min(COB, (p.Cout - co0 + 31) / 32), nci = min(NCI, (p.Cin - ci0 + 31) / 32);

    __shared__ float gyT[2][NCO * WG_LDA];
    __shared__ float xT[2][NCX * WG_LDA];
    f32x16 acc[COB][NCI];
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < NCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool vec = ((p.T & 3) == 0) && ((p.gy.sB | p.gy.sC | p.gy.sF | p.x.sB | p.x.sC | p.x.sF) & 3) == 0 &&
                     ((((uintptr_t)p.gy.p) | ((uintptr_t)p.x.p)) & 15) == 0;
    auto ld4 = [&](const float* base, bool ok, int t) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && t < p.T) {
            const float* src = base + t;
            if (vec && t + 3 < p.T) v = *reinterpret_cast<const float4*>(src);
            else { v.x = src[0]; if (t + 1 < p.T) v.y = src[1]; if (t + 2 < p.T) v.z = src[2]; if (t + 3 < p.T) v.w = src[3]; }
        }
        return v;
    };
    const int q4 = tid & 15, srow = tid >> 4;             // float4 column, base row 0..31 (rows srow + 32 i)
    auto load = [&](int f, int t0, float4 (&g)[COB], float4 (&x)[NCI]) {
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            const int co = co0 + srow + 32 * i;
            g[i] = ld4(p.gy.p + (int64_t)b * p.gy.sB + (int64_t)co * p.gy.sC + (int64_t)f * p.gy.sF, co < p.Cout, t0 + 4 * q4);
        }
#pragma unroll
        for (int j = 0; j < NCI; ++j) {
            const int ci = ci0 + srow + 32 * j;
            x[j] = ld4(p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)f * p.x.sF, ci < p.Cin, t0 + 4 * q4);
        }
    };
    auto store = [&](int buf, const float4 (&g)[COB], const float4 (&x)[NCI]) {
#pragma unroll
        for (int i = 0; i < COB; ++i) {
            float* d = gyT[buf] + (srow + 32 * i) * WG_LDA + 4 * q4;
            d[0] = g[i].x; d[1] = g[i].y; d[2] = g[i].z; d[3] = g[i].w;
        }
#pragma unroll
        for (int j = 0; j < NCI; ++j) {
            float* d = xT[buf] + (srow + 32 * j) * WG_LDA + 4 * q4;
            d[0] = x[j].x; d[1] = x[j].y; d[2] = x[j].z; d[3] = x[j].w;
        }
    };
    // steps: (chunk of 64 positions, row), rows fastest; the S splits of a sample cut the sequence into equal contiguous parts
    const int nct = (p.T + WG_TC - 1) / WG_TC;
    const int tc = (p.T < WG_TC) ? ((p.T + 3) & ~3) : WG_TC;
    const int64_t total = (int64_t)nct * p.F;
    const int g_lo = (int)((total * s) / p.S), g_hi = (int)((total * (s + 1)) / p.S);
    float4 pg[COB], px[NCI];
    if (g_lo < g_hi) {
        load(g_lo % p.F, (g_lo / p.F) * WG_TC, pg, px);
        store(0, pg, px);
    }
    __syncthreads();
    int cur = 0;
    for (int g = g_lo; g < g_hi; ++g) {
        const bool have = g + 1 < g_hi;
        if (have) load((g + 1) % p.F, ((g + 1) / p.F) * WG_TC, pg, px);
        const float* ap = gyT[cur] + l32 * WG_LDA + half;
        const float* bp = xT[cur] + l32 * WG_LDA + half;
        bool stored = !have;
        for (int k = 2 * wave; k < tc; k += 2 * WG_NW) {
            if (!stored && k >= (tc >> 1)) { store(cur ^ 1, pg, px); stored = true; }
            float av[COB], bv[NCI];
#pragma unroll
            for (int i = 0; i < COB; ++i) av[i] = (i < nco) ? ap[i * 32 * WG_LDA + k] : 0.f;
#pragma unroll
            for (int j = 0; j < NCI; ++j) bv[j] = (j < nci) ? bp[j * 32 * WG_LDA + k] : 0.f;
#pragma unroll
            for (int i = 0; i < COB; ++i)
#pragma unroll
                for (int j = 0; j < NCI; ++j)
                    if (i < nco && j < nci) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (!stored) store(cur ^ 1, pg, px);
        __syncthreads();
        cur ^= 1;
    }
    // fixed-order sum of the eight waves' tiles through LDS, then P[(b*S+s)][co][ci]
    float* red = gyT[0];                                    // [wave][16][64] floats = 32 KB (NCO >= 64: 2 x 64 x 65 floats are there)
    float* P = p.P + ((int64_t)(b * p.S + s) * p.Cout) * p.Cin;
#pragma unroll
    for (int i = 0; i < COB; ++i)
#pragma unroll
        for (int j = 0; j < NCI; ++j) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[i][j][r];
            __syncthreads();
            if (wave == 0) {
                const int ci = ci0 + 32 * j + l32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = red[r * 64 + lane];
                    for (int w = 1; w < WG_NW; ++w) v += red[(w * 16 + r) * 64 + lane];
                    const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout && ci < p.Cin) P[(int64_t)co * p.Cin + ci] = p.alpha * v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------
// Winograd F(4,3) form of the weight gradient (5x3 layers): with y = A^T[(G w) . (B^T d)] per group of four outputs,
//   dL/dU[co,ci,kh,xi] = sum_{b,f,g} (A gy)[co,f,g,xi] * (B^T d)[ci,f+(kh-2)dil,g,xi],      dL/dw = G^T dL/dU   (aid_wgrad_reduce),
// i.e. 6 products per 4 positions and tap row instead of 12: half the MFMAs of the direct form.  Both operands arrive in the Winograd
// domain [.,.,F,6,T/4] (gy: aid_wino_gy, d: aid_scale_act(wino=1), so no halo samples); a step is one output row x 16 groups; the 30
// (xi,kh) pairs of a layer are spread over the 8 waves (pairs wave, wave+8, wave+16, wave+24); staging, ring and step sequence as in
// conv_wgrad_kernel.
#define WW_G 16                  // groups (of 4 samples) per staged chunk
#define WW_ROW (6 * WW_G + 1)    // one channel's [xi][g] block in LDS, padded to an odd length (conflict-free transposed fragment reads)
#define WW_QPW 4                 // pairs per wave
__global__ __launch_bounds__(WG_THREADS, 2) void conv_wgrad_wino_kernel(const WgDev a) {
    constexpr int COB = 2, WG_CO = 64, KH = 5, NS = KH + 1, NPAIR = 30;
    const aid_conv2d_wgrad_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = (blockIdx.x % a.co_tiles) * WG_CO;
    const int ci0 = (blockIdx.x / a.co_tiles) * 32;
    const int b = blockIdx.y / p.S, s = blockIdx.y - b * p.S;
    const int nco = min(COB, (p.Cout - co0 + 31) / 32);
    const int G = p.T >> 2;                               // groups per row

    __shared__ float gyT0[WG_CO * WW_ROW + 8];
    __shared__ float gyT1[WG_CO * WW_ROW + 8];
    __shared__ float xT[NS * 32 * WW_ROW + 8];

    f32x16 acc[WW_QPW][COB];
#pragma unroll
    for (int q = 0; q < WW_QPW; ++q)
#pragma unroll
        for (int i = 0; i < COB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;
    int pkh[WW_QPW], pxi[WW_QPW];                         // pair = xi * 5 + kh (the Winograd pack's tap order)
#pragma unroll
    for (int q = 0; q < WW_QPW; ++q) { const int pr = min(wave + WG_NW * q, NPAIR - 1); pxi[q] = pr / KH; pkh[q] = pr - pxi[q] * KH; }

    // staging shares: gy 64 channels x 6 planes x 4 float4 = 1536 float4 -> 3 per thread; one input row 32 x 24 = 768 float4 -> threads < 256 take 2
    auto gy_src = [&](int e, int f, int g0) {             // e: float4 index inside the tile, [co][xi][4]
        const int co = co0 + e / 24, r = e % 24, xi = r >> 2, q4 = r & 3;
        const int g = g0 + 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < p.Cout && g < G) v = *reinterpret_cast<const float4*>(p.gy.p + (int64_t)b * p.gy.sB + (int64_t)co * p.gy.sC + (int64_t)f * p.gy.sF + (int64_t)xi * G + g);
        return v;
    };
    auto x_src = [&](int e, int fi, int g0) {
        const int ci = ci0 + e / 24, r = e % 24, xi = r >> 2, q4 = r & 3;
        const int g = g0 + 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < p.Cin && fi >= 0 && fi < p.F && g < G) v = *reinterpret_cast<const float4*>(p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF + (int64_t)xi * G + g);
        return v;
    };
    auto lds_put = [&](float* base, int e, const float4& v) {  // [ch][xi][g16] with the odd channel pitch
        const int ch = e / 24, r = e % 24;
        float* d = base + ch * WW_ROW + (r >> 2) * WW_G + 4 * (r & 3);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    };
    auto load_gy = [&](int f, int g0, float4 (&g)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = gy_src(tid + WG_THREADS * i, f, g0);
    };
    auto store_gy = [&](float* gyT, const float4 (&g)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) lds_put(gyT, tid + WG_THREADS * i, g[i]);
    };
    auto load_row = [&](int fi, int g0, float4 (&x)[2]) {
        x[0] = x_src(tid, fi, g0);
        x[1] = (tid < 768 - WG_THREADS) ? x_src(tid + WG_THREADS, fi, g0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_row = [&](int slot, const float4 (&x)[2]) {
        lds_put(xT + slot * 32 * WW_ROW, tid, x[0]);
        if (tid < 768 - WG_THREADS) lds_put(xT + slot * 32 * WW_ROW, tid + WG_THREADS, x[1]);
    };

    const int dil = p.dilF;
    const int nct = (G + WW_G - 1) / WW_G;
    const int gc = (G < WW_G) ? ((G + 1) & ~1) : WW_G;         // groups walked per chunk (T = 32: 8)
    const int qd = p.F / dil, rem = p.F - qd * dil;
    const int64_t total = (int64_t)nct * p.F;
    const int g_lo = (int)((total * s) / p.S), g_hi = (int)((total * (s + 1)) / p.S);
    int c = g_lo / p.F, res, n;
    {
        const int r = g_lo - c * p.F;
        if (r < rem * (qd + 1)) { res = r / (qd + 1); n = r - res * (qd + 1); }
        else { const int r2 = r - rem * (qd + 1); res = rem + r2 / qd; n = r2 - (res - rem) * qd; }
    }
    float4 pg[3], px[2];
    auto stage_chain_start = [&](int f, int g0, int n0) {
        for (int kh = 0; kh < KH; ++kh) {
            load_row(f + (kh - 2) * dil, g0, px);
            store_row((n0 + kh) % NS, px);
        }
    };
    if (g_lo < g_hi) {
        load_gy(res + n * dil, c * WW_G, pg);
        store_gy(gyT0, pg);
        stage_chain_start(res + n * dil, c * WW_G, n);
    }
    __syncthreads();
    int cur = 0;
    for (int g = g_lo; g < g_hi; ++g) {
        int n2 = n + 1, res2 = res, c2 = c;
        if (n2 >= qd + (res < rem ? 1 : 0)) { n2 = 0; ++res2; if (res2 >= dil || res2 >= p.F) { res2 = 0; ++c2; } }
        const bool have = g + 1 < g_hi;
        const bool chain = n2 != 0;
        const int f2 = res2 + n2 * dil;
        if (have) {
            load_gy(f2, c2 * WW_G, pg);
            if (chain) load_row(f2 + (KH - 1 - 2) * dil, c2 * WW_G, px);
        }
        const float* gyT = cur ? gyT1 : gyT0;
        float* gyN = cur ? gyT0 : gyT1;
        // fragment bases: A = (A gy)[co = l32 (+32 i)][xi][g = 2 k + half], B = V[slot n + kh][ci = l32][xi][g = 2 k + half]
        int aoff[WW_QPW], boff[WW_QPW];
#pragma unroll
        for (int q = 0; q < WW_QPW; ++q) {
            aoff[q] = l32 * WW_ROW + pxi[q] * WW_G + half;
            boff[q] = (((n + pkh[q]) % NS) * 32 + l32) * WW_ROW + pxi[q] * WW_G + half;
        }
        const int kmid = (gc >> 2) << 1;
        for (int k = 0; k < gc; k += 2) {
            if (k == kmid && have) {
                store_gy(gyN, pg);
                if (chain) store_row((n2 + KH - 1) % NS, px);
            }
            float av[WW_QPW][COB], bv[WW_QPW];
#pragma unroll
            for (int q = 0; q < WW_QPW; ++q) {
                bv[q] = xT[boff[q] + k];
#pragma unroll
                for (int i = 0; i < COB; ++i) av[q][i] = (i < nco) ? gyT[aoff[q] + i * 32 * WW_ROW + k] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < WW_QPW; ++q)
#pragma unroll
                for (int i = 0; i < COB; ++i)
                    if (i < nco) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], bv[q], acc[q][i], 0, 0, 0);
        }
        __syncthreads();
        if (have && !chain) {
            stage_chain_start(f2, c2 * WW_G, n2);
            __syncthreads();
        }
        cur ^= 1;
        n = n2; res = res2; c = c2;
    }
    // ---- partials P[(b*S+s)][co][pair (30)][ci] -------------------------------------------------------------------------------------
    float* P = p.P + ((int64_t)(b * p.S + s) * NPAIR) * p.Cout * p.Cin;
    const int ci = ci0 + l32;
#pragma unroll
    for (int q = 0; q < WW_QPW; ++q) {
        const int pr = wave + WG_NW * q;
        if (pr >= NPAIR || ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < COB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.Cout) P[((int64_t)co * NPAIR + pr) * p.Cin + ci] = p.alpha * acc[q][i][r];
            }
    }
}

// (A gy) per group of four samples: [B,C,F,T] -> [B,C,F,6,T/4]   (A = transpose of the F(4,3) output transform; no neighbour samples)
__global__ __launch_bounds__(256) void wino_gy_kernel(const aid_wino_gy_params p) {
    const int G = p.T >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.B * p.C * p.F * G) return;
    const int g = (int)(i % G);
    const int64_t row = i / G;
    const int f = (int)(row % p.F);
    const int64_t bc = row / p.F;
    const int c = (int)(bc % p.C), b = (int)(bc / p.C);
    const float4 v = *reinterpret_cast<const float4*>(p.gy.p + (int64_t)b * p.gy.sB + (int64_t)c * p.gy.sC + (int64_t)f * p.gy.sF + 4 * g);
    float* o = p.out.p + (int64_t)b * p.out.sB + (int64_t)c * p.out.sC + (int64_t)f * p.out.sF + g;
    const float s02 = v.x + v.z, s13 = v.y + v.w, s04 = v.x + 4.f * v.z, s28 = 2.f * v.y + 8.f * v.w;
    o[0] = v.x;
    o[G] = s02 + s13;
    o[2 * G] = s02 - s13;
    o[3 * G] = s04 + s28;
    o[4 * G] = s04 - s28;
    o[5 * G] = v.w;
}

extern "C" int aid_wino_gy(const aid_wino_gy_params* p, void* stream) {
    AID_REQUIRE(p && p->gy.p && p->out.p, "aid_wino_gy: null pointer");
    AID_REQUIRE(p->B > 0 && p->C > 0 && p->F > 0 && p->T > 0 && (p->T % 4) == 0, "aid_wino_gy: T % 4 == 0");
    AID_REQUIRE(((p->gy.sB | p->gy.sC | p->gy.sF) & 3) == 0 && (((uintptr_t)p->gy.p) & 15) == 0, "aid_wino_gy: gy must be float4-addressable");
    AID_REQUIRE(p->out.sF >= 6 * (p->T / 4), "aid_wino_gy: out rows are [6][T/4]");
    const int64_t n = (int64_t)p->B * p->C * p->F * (p->T / 4);
    hipLaunchKernelGGL(wino_gy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// workgroups per (sample, split) that aid_conv2d_wgrad launches for a layer: the caller sizes S with it (about one workgroup per CU in total)
extern "C" int aid_conv2d_wgrad_tiles(int Cin, int Cout, int KH, int KW, int wino) {
    if (wino) return aid_cdiv(Cout, 64) * aid_cdiv(Cin, 32);
    if (KH == 1 && KW == 1 && Cout >= 64 && Cin >= 64) return aid_cdiv(Cout, Cout >= 128 ? 128 : 64) * aid_cdiv(Cin, 64);
    const int cob = (Cout % 64 != 0 && Cout % 96 == 0) ? 3 : 2;
    return aid_cdiv(Cout, 32 * cob) * aid_cdiv(Cin, 32);
}

extern "C" int aid_conv2d_wgrad(const aid_conv2d_wgrad_params* p, void* stream) {
    AID_REQUIRE(p && p->gy.p && p->x.p && p->P, "aid_conv2d_wgrad: null pointer");
    AID_REQUIRE(p->B > 0 && p->Cin > 0 && p->Cout > 0 && p->F > 0 && p->T > 0 && p->S >= 1 && p->S <= p->F, "aid_conv2d_wgrad: bad shape");
    AID_REQUIRE((int64_t)p->F * aid_cdiv(p->T, WG_TC) < (1LL << 30), "aid_conv2d_wgrad: too many steps");
    AID_REQUIRE(p->KH >= 1 && p->KH <= WG_MAXKH && (p->KW == 1 || p->KW == 3) && p->KH * p->KW <= WG_MAXTAPS && p->KH * p->KW <= 16,
                "aid_conv2d_wgrad: kernel sizes up to 5x3");
    WgDev a;
    a.p = *p;
    if (p->wino) {
        AID_REQUIRE(p->KH == 5 && p->KW == 3 && (p->T % 8) == 0 && p->dilF >= 1, "aid_conv2d_wgrad: the F(4,3) form is for 5x3 layers with T % 8 == 0");
        AID_REQUIRE(p->gy.sF >= 6 * (p->T / 4) && p->x.sF >= 6 * (p->T / 4), "aid_conv2d_wgrad: wino operands have rows [6][T/4]");
        AID_REQUIRE(((p->gy.sB | p->gy.sC | p->gy.sF | p->x.sB | p->x.sC | p->x.sF) & 3) == 0 && ((((uintptr_t)p->gy.p) | ((uintptr_t)p->x.p)) & 15) == 0 && (p->T % 16) == 0,
                    "aid_conv2d_wgrad: wino operands must be float4-addressable (T % 16 == 0)");
        a.co_tiles = aid_cdiv(p->Cout, 64);
        a.ci_tiles = aid_cdiv(p->Cin, 32);
        AID_REQUIRE((int64_t)p->B * p->S < 65536, "aid_conv2d_wgrad: too many (sample, split) pairs");
        hipLaunchKernelGGL(conv_wgrad_wino_kernel, dim3((unsigned)(a.co_tiles * a.ci_tiles), (unsigned)(p->B * p->S)), dim3(WG_THREADS), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    if (p->KH == 1 && p->KW == 1 && p->Cout >= 64 && p->Cin >= 64) {           // 1x1: position-split GEMM blocks of 128 x 64 or 64 x 64
        const bool big = p->Cout >= 128;
        a.co_tiles = aid_cdiv(p->Cout, big ? 128 : 64);
        a.ci_tiles = aid_cdiv(p->Cin, 64);
        AID_REQUIRE((int64_t)p->B * p->S < 65536, "aid_conv2d_wgrad: too many (sample, split) pairs");
        const dim3 grid1((unsigned)(a.co_tiles * a.ci_tiles), (unsigned)(p->B * p->S));
        if (big) hipLaunchKernelGGL((conv_wgrad_1x1_kernel<4, 2>), grid1, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_wgrad_1x1_kernel<2, 2>), grid1, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
        AID_CHECK_LAUNCH();
        return AID_OK;
    }
    const int cob = (p->Cout % 64 != 0 && p->Cout % 96 == 0) ? 3 : 2;      // 96-channel tiles for C = 96 (no half-empty second tile)
    a.co_tiles = aid_cdiv(p->Cout, 32 * cob);
    a.ci_tiles = aid_cdiv(p->Cin, 32);
    AID_REQUIRE((int64_t)p->B * p->S < 65536, "aid_conv2d_wgrad: too many (sample, split) pairs");
    const dim3 grid((unsigned)(a.co_tiles * a.ci_tiles), (unsigned)(p->B * p->S));
    if (cob == 3) hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(WG_THREADS), 0, (hipStream_t)stream, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_reduce_w_kernel(const aid_wgrad_reduce_params p) {
    const int64_t n = (int64_t)p.Cout * p.Cin * p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // index into one partial: [co][tap][ci] (coalesced reads)
    if (i >= n) return;
    const int ci = (int)(i % p.Cin);
    const int tap = (int)((i / p.Cin) % p.K);
    const int co = (int)(i / ((int64_t)p.Cin * p.K));
    float accv = 0.f;
    if (p.wino) {                                               // P: [co][xi*5+kh][ci] (30 taps); dW[kh][kw] = sum_xi G[xi][kw] dU[xi][kh]
        // one thread per (co, kh, ci): the six xi values are read once and give all three kw (threads with tap % 3 != 0 have nothing to do)
        const int kh = tap / 3;
        if (tap - 3 * kh != 0) return;
        const int64_t n30 = (int64_t)p.Cout * p.Cin * 30;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int b = 0; b < p.B; ++b) {
            float u[6];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                float sp = 0.f;
                for (int s = 0; s < p.S; ++s) sp += p.P[((int64_t)(b * p.S + s)) * n30 + ((int64_t)co * 30 + xi * 5 + kh) * p.Cin + ci];
                u[xi] = sp;
            }
            const float g = (p.gate ? p.gate[(int64_t)b * p.gate_ld + co] : 1.f) * (p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_ld + ci] : 1.f);
            a0 += g * (0.25f * u[0] - (u[1] + u[2]) * (1.f / 6) + (u[3] + u[4]) * (1.f / 24));
            a1 += g * ((u[2] - u[1]) * (1.f / 6) + (u[3] - u[4]) * (1.f / 12));
            a2 += g * (u[5] - (u[1] + u[2]) * (1.f / 6) + (u[3] + u[4]) * (1.f / 6));
        }
        const int64_t o = ((int64_t)co * p.Cin + ci) * p.K + 3 * kh;
        p.dW[o] = (p.accumulate ? p.dW[o] : 0.f) + a0;
        p.dW[o + 1] = (p.accumulate ? p.dW[o + 1] : 0.f) + a1;
        p.dW[o + 2] = (p.accumulate ? p.dW[o + 2] : 0.f) + a2;
        return;
    } else {
        for (int b = 0; b < p.B; ++b) {                         // fixed order: deterministic
            float sp = 0.f;
            for (int s = 0; s < p.S; ++s) sp += p.P[((int64_t)(b * p.S + s)) * n + i];
            const float g = p.gate ? p.gate[(int64_t)b * p.gate_ld + co] : 1.f;
            const float sc = p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_ld + ci] : 1.f;
            accv += g * sc * sp;
        }
    }
    const int64_t o = ((int64_t)co * p.Cin + ci) * p.K + tap;       // the parameter's layout [co][ci][tap]
    p.dW[o] = (p.accumulate ? p.dW[o] : 0.f) + accv;
}

#define WRG_CHUNK 3840            // floats of a weight row staged per pass (256 input channels x 15 taps)
__global__ __launch_bounds__(256) void wgrad_reduce_gate_kernel(const aid_wgrad_reduce_params p) {
    const int b = blockIdx.y, co = blockIdx.x, tid = threadIdx.x;
    const int64_t row = (int64_t)p.Cin * p.K;
    const int64_t n = (int64_t)p.Cout * row;
    __shared__ float Wl[WRG_CHUNK];
    __shared__ double red[4];
    double accv = 0.0;
    const int cpc = WRG_CHUNK / p.K;                        // input channels per staged chunk
    if (p.wino) {                                           // <W, G^T dU> = <G W, dU>: the U-domain partials against the layer's F(4,3) weight pack
        const int64_t n30 = (int64_t)p.Cout * p.Cin * 30;
        for (int j = tid; j < 30 * p.Cin; j += 256) {       // j = pair * Cin + ci (the partials' order)
            const int pr = j / p.Cin, ci = j - pr * p.Cin;
            float sp = 0.f;
            for (int s = 0; s < p.S; ++s) sp += p.P[((int64_t)(b * p.S + s)) * n30 + (int64_t)co * 30 * p.Cin + j];
            const float sc = p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_ld + ci] : 1.f;
            accv += (double)(p.Uw[((int64_t)pr * p.Cin_pad + ci) * p.Cout_pad + co] * sc) * (double)sp;
        }
    } else
    for (int c0 = 0; c0 < p.Cin; c0 += cpc) {
        const int nc = min(cpc, p.Cin - c0);
        __syncthreads();
        for (int e = tid; e < nc * p.K; e += 256) Wl[e] = p.W[(int64_t)co * row + (int64_t)c0 * p.K + e];     // [ci][tap], coalesced
        __syncthreads();
        for (int j = tid; j < nc * p.K; j += 256) {         // j = tap * nc + cl: the partials' order ([co][tap][ci]) -> coalesced reads
            const int tap = j / nc, cl = j - tap * nc, ci = c0 + cl;
            float sp = 0.f;
            for (int s = 0; s < p.S; ++s) sp += p.P[((int64_t)(b * p.S + s)) * n + (int64_t)co * row + (int64_t)tap * p.Cin + ci];
            const float sc = p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_ld + ci] : 1.f;
            accv += (double)(Wl[cl * p.K + tap] * sc) * (double)sp;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) accv += __shfl_down(accv, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = accv;
    __syncthreads();
    if (tid == 0) p.dgate[(int64_t)b * p.dgate_ld + co] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int aid_wgrad_reduce(const aid_wgrad_reduce_params* p, void* stream) {
    AID_REQUIRE(p && p->P && p->dW, "aid_wgrad_reduce: null pointer");
    AID_REQUIRE(!p->dgate || p->W, "aid_wgrad_reduce: dgate needs the weights");
    AID_REQUIRE(p->K >= 1 && p->K <= WRG_CHUNK, "aid_wgrad_reduce: bad tap count");
    AID_REQUIRE(!p->wino || (p->K == 15 && (!p->dgate || (p->Uw && p->Cin_pad >= p->Cin && p->Cout_pad >= p->Cout))), "aid_wgrad_reduce: wino needs K = 15 (5x3) and, for dgate, the F(4,3) weight pack");
    const int64_t n = (int64_t)p->Cout * p->Cin * p->K;
    hipLaunchKernelGGL(wgrad_reduce_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    if (p->dgate) {
        hipLaunchKernelGGL(wgrad_reduce_gate_kernel, dim3((unsigned)p->Cout, (unsigned)p->B), dim3(256), 0, (hipStream_t)stream, *p);
        AID_CHECK_LAUNCH();
    }
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_dot_kernel(const aid_channel_dot_params p, int vec) {
    const int bc = blockIdx.x, b = bc / p.C, c = bc - b * p.C;
    const int tid = threadIdx.x;
    const float* u = p.u.p + (int64_t)b * p.u.sB + (int64_t)c * p.u.sC;
    const float* v = p.v.p + (int64_t)b * p.v.sB + (int64_t)c * p.v.sC;
    double s = 0.0;
    if (vec) {                                              // rows of float4 (T % 4 == 0, 16-byte aligned views)
        const int T4 = p.T >> 2;
        const int n4 = p.F * T4;
        for (int i = tid; i < n4; i += 256) {
            const int f = i / T4, t = (i - f * T4) << 2;
            const float4 a = *reinterpret_cast<const float4*>(u + (int64_t)f * p.u.sF + t);
            const float4 w = *reinterpret_cast<const float4*>(v + (int64_t)f * p.v.sF + t);
            s += (double)(a.x * w.x + a.y * w.y) + (double)(a.z * w.z + a.w * w.w);
        }
    } else {
        const int64_t n = (int64_t)p.F * p.T;
        for (int64_t i = tid; i < n; i += 256) {
            const int f = (int)(i / p.T), t = (int)(i - (int64_t)f * p.T);
            s += (double)u[(int64_t)f * p.u.sF + t] * (double)v[(int64_t)f * p.v.sF + t];
        }
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) p.out[(int64_t)b * p.out_ld + c] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// Gradient of the relative-position embedding (attention_dict.use_rel_pos; RelativePositionBias, unet...py:266-312): the additive logit bias is
// table[h][n][m] = W[bucket[n][m]][h], so dW[k][h] = sum over samples and over the (n, m) of bucket k of dS[b][h][n][m] -- dS as
// aid_time_attention_bwd leaves it in its scratch (the gradient w.r.t. q k^T + bias, i.e. before the scale).  One workgroup per (bucket, head),
// fixed-order reduction (fp64), no atomics.
__global__ __launch_bounds__(256) void relpos_bwd_kernel(const aid_relpos_bwd_params p) {
    const int k = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int n2 = p.T * p.T;
    double s = 0.0;
    for (int i = tid; i < n2; i += 256) {
        if (p.bucket[i] != k) continue;
        for (int b = 0; b < p.B; ++b) s += (double)p.dS[((int64_t)b * p.H + h) * n2 + i];
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float v = (float)((red[0] + red[1]) + (red[2] + red[3]));
        float* o = p.dW + (int64_t)k * p.H + h;
        *o = p.accumulate ? *o + v : v;
    }
}

extern "C" int aid_relpos_bwd(const aid_relpos_bwd_params* p, void* stream) {
    AID_REQUIRE(p && p->dS && p->bucket && p->dW, "aid_relpos_bwd: null pointer");
    AID_REQUIRE(p->B > 0 && p->H > 0 && p->T > 0 && p->num_buckets > 0, "aid_relpos_bwd: empty shape");
    hipLaunchKernelGGL(relpos_bwd_kernel, dim3((unsigned)p->num_buckets, (unsigned)p->H), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

extern "C" int aid_channel_dot(const aid_channel_dot_params* p, void* stream) {
    AID_REQUIRE(p && p->u.p && p->v.p && p->out, "aid_channel_dot: null pointer");
    const int vec = ((p->T & 3) == 0) && ((p->u.sB | p->u.sC | p->u.sF | p->v.sB | p->v.sC | p->v.sF) & 3) == 0 &&
                    ((((uintptr_t)p->u.p) | ((uintptr_t)p->v.p)) & 15) == 0;
    hipLaunchKernelGGL(channel_dot_kernel, dim3((unsigned)(p->B * p->C)), dim3(256), 0, (hipStream_t)stream, *p, vec);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// d scale[b,c] = S[b,c] / scale[b,c];  scale = gamma_c (1 + a[b,c]) inv[b,g]
__global__ __launch_bounds__(256) void scale_bwd_kernel(const aid_scale_bwd_params p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    const int g = c / (p.C / p.groups);
    const float gam = p.gamma[c];
    float dg = 0.f;
    for (int b = 0; b < p.B; ++b) {
        const float sc = p.scale[(int64_t)b * p.scale_ld + c];
        const float ds = (sc != 0.f) ? p.S[(int64_t)b * p.S_ld + c] / sc : 0.f;
        const float inv = p.stats[((int64_t)b * p.groups + g) * 2 + 1];
        const float a1 = 1.f + (p.mod ? p.mod[(int64_t)b * p.mod_ld + c] : 0.f);
        dg += ds * a1 * inv;
        if (p.dmod) p.dmod[(int64_t)b * p.dmod_ld + c] = ds * gam * inv;
    }
    p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + dg;
}

extern "C" int aid_scale_bwd(const aid_scale_bwd_params* p, void* stream) {
    AID_REQUIRE(p && p->S && p->scale && p->gamma && p->stats && p->dgamma, "aid_scale_bwd: null pointer");
    AID_REQUIRE(p->groups > 0 && (p->C % p->groups) == 0, "aid_scale_bwd: C must be a multiple of groups");
    hipLaunchKernelGGL(scale_bwd_kernel, dim3((unsigned)aid_cdiv(p->C, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
// mod[b,j] = emb[b,:] . W[j,:] + bias[j]   ->   dW[j,e] = sum_b dmod[b,j] emb[b,e],  dbias[j] = sum_b dmod[b,j],
//                                              demb[b,e] = sum_j dmod[b,j] W[j,e]
__global__ __launch_bounds__(256) void modulation_bwd_w_kernel(const aid_modulation_bwd_params p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.N * p.E) return;
    const int j = (int)(i / p.E), e = (int)(i - (int64_t)j * p.E);
    float accv = 0.f;
    for (int b = 0; b < p.B; ++b) accv += p.dmod[(int64_t)b * p.N + j] * p.emb[(int64_t)b * p.E + e];
    p.dW[i] = (p.accumulate ? p.dW[i] : 0.f) + accv;
    if (e == 0) {
        float sb = 0.f;
        for (int b = 0; b < p.B; ++b) sb += p.dmod[(int64_t)b * p.N + j];
        p.dbias[j] = (p.accumulate ? p.dbias[j] : 0.f) + sb;
    }
}

// demb[b,e] = sum_j dmod[b,j] W[j,e]: one workgroup per (row chunk of 128 j, sample), partial sums to `part`, then a fixed-order fold
#define MODB_CH 128
__global__ __launch_bounds__(256) void modulation_bwd_e_kernel(const aid_modulation_bwd_params p, float* __restrict__ part, int nchunk) {
    const int b = blockIdx.y, ch = blockIdx.x;
    const int j0 = ch * MODB_CH, j1 = min(p.N, j0 + MODB_CH);
    for (int e = threadIdx.x; e < p.E; e += 256) {
        float accv = 0.f;
        for (int j = j0; j < j1; ++j) accv += p.dmod[(int64_t)b * p.N + j] * p.W[(int64_t)j * p.E + e];
        part[((int64_t)b * nchunk + ch) * p.E + e] = accv;
    }
}

__global__ __launch_bounds__(256) void modulation_bwd_e_fold_kernel(const aid_modulation_bwd_params p, const float* __restrict__ part, int nchunk) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.E) return;
    float accv = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) accv += part[((int64_t)b * nchunk + ch) * p.E + e];
    p.demb[(int64_t)b * p.E + e] = accv;
}

extern "C" int aid_modulation_bwd(const aid_modulation_bwd_params* p, void* stream) {
    AID_REQUIRE(p && p->dmod && p->emb && p->W && p->dW && p->dbias && p->demb, "aid_modulation_bwd: null pointer");
    hipLaunchKernelGGL(modulation_bwd_w_kernel, dim3((unsigned)(((int64_t)p->N * p->E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    const int nchunk = aid_cdiv(p->N, MODB_CH);
    AID_REQUIRE(p->part && p->part_floats >= (int64_t)p->B * nchunk * p->E, "aid_modulation_bwd: `part` scratch must hold B * ceil(N/128) * E floats");
    hipLaunchKernelGGL(modulation_bwd_e_kernel, dim3((unsigned)nchunk, (unsigned)p->B), dim3(256), 0, (hipStream_t)stream, *p, p->part, nchunk);
    AID_CHECK_LAUNCH();
    hipLaunchKernelGGL(modulation_bwd_e_fold_kernel, dim3((unsigned)aid_cdiv(p->E, 256), (unsigned)p->B), dim3(256), 0, (hipStream_t)stream, *p,
                       (const float*)p->part, nchunk);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// RFF-MLP backward: one workgroup walks the samples in order (deterministic accumulation); the forward activations are
// recomputed per sample in LDS exactly as aid_embed computes them.
#define EMB_MAX 1024
__global__ __launch_bounds__(256) void embed_bwd_kernel(const aid_embed_bwd_params q) {
    const aid_embed_params& p = q.fwd;
    __shared__ float x0[EMB_MAX], a0[EMB_MAX], a1[EMB_MAX], g2[EMB_MAX], g1[EMB_MAX], g0[EMB_MAX];
    const int tid = threadIdx.x;
    const int d0 = 2 * p.rff;
    for (int b = 0; b < p.B; ++b) {
        const float s = p.sigma[b];
        for (int i = tid; i < p.rff; i += 256) {
            const float ang = 2.0f * 3.14159265358979323846f * s * p.rff_freq[i];
            x0[i] = sinf(ang);
            x0[p.rff + i] = cosf(ang);
        }
        __syncthreads();
        for (int o = tid; o < p.h0; o += 256) {
            float accv = p.b0[o];
            for (int i = 0; i < d0; ++i) accv += x0[i] * p.w0[(int64_t)o * d0 + i];
            a0[o] = fmaxf(accv, 0.f);
        }
        __syncthreads();
        for (int o = tid; o < p.h1; o += 256) {
            float accv = p.b1[o];
            for (int i = 0; i < p.h0; ++i) accv += a0[i] * p.w1[(int64_t)o * p.h0 + i];
            a1[o] = fmaxf(accv, 0.f);
        }
        __syncthreads();
        // layer 2 (output E): the forward output is emb = relu(z2); g2 = demb * (emb > 0)
        for (int o = tid; o < p.E; o += 256) g2[o] = (p.emb[(int64_t)b * p.E + o] > 0.f) ? q.demb[(int64_t)b * p.E + o] : 0.f;
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)p.E * p.h1; i += 256) q.dw2[i] = ((b || q.accumulate) ? q.dw2[i] : 0.f) + g2[i / p.h1] * a1[i % p.h1];
        for (int o = tid; o < p.E; o += 256) q.db2[o] = ((b || q.accumulate) ? q.db2[o] : 0.f) + g2[o];
        for (int i = tid; i < p.h1; i += 256) {
            float accv = 0.f;
            for (int o = 0; o < p.E; ++o) accv += g2[o] * p.w2[(int64_t)o * p.h1 + i];
            g1[i] = (a1[i] > 0.f) ? accv : 0.f;
        }
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)p.h1 * p.h0; i += 256) q.dw1[i] = ((b || q.accumulate) ? q.dw1[i] : 0.f) + g1[i / p.h0] * a0[i % p.h0];
        for (int o = tid; o < p.h1; o += 256) q.db1[o] = ((b || q.accumulate) ? q.db1[o] : 0.f) + g1[o];
        for (int i = tid; i < p.h0; i += 256) {
            float accv = 0.f;
            for (int o = 0; o < p.h1; ++o) accv += g1[o] * p.w1[(int64_t)o * p.h0 + i];
            g0[i] = (a0[i] > 0.f) ? accv : 0.f;
        }
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)p.h0 * d0; i += 256) q.dw0[i] = ((b || q.accumulate) ? q.dw0[i] : 0.f) + g0[i / d0] * x0[i % d0];
        for (int o = tid; o < p.h0; o += 256) q.db0[o] = ((b || q.accumulate) ? q.db0[o] : 0.f) + g0[o];
        __syncthreads();
    }
}

extern "C" int aid_embed_bwd(const aid_embed_bwd_params* q, void* stream) {
    AID_REQUIRE(q && q->demb && q->dw0 && q->dw1 && q->dw2 && q->db0 && q->db1 && q->db2 && q->fwd.emb, "aid_embed_bwd: null pointer");
    AID_REQUIRE(2 * q->fwd.rff <= EMB_MAX && q->fwd.h0 <= EMB_MAX && q->fwd.h1 <= EMB_MAX && q->fwd.E <= EMB_MAX, "aid_embed_bwd: layer too wide");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *q);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
// torch.optim.Adam (no weight decay, no amsgrad): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          gscale: gradient-clipping coefficient (1 = none)
__global__ __launch_bounds__(256) void adam_kernel(const aid_adam_params p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float g = p.grad[i] * (p.gscale ? *p.gscale : 1.f);
    const float m = p.beta1 * p.m[i] + (1.f - p.beta1) * g;
    const float v = p.beta2 * p.v[i] + (1.f - p.beta2) * g * g;
    p.m[i] = m;
    p.v[i] = v;
    const float denom = sqrtf(v) / p.bias2_sqrt + p.eps;
    p.param[i] -= (p.lr / p.bias1) * (m / denom);
}

extern "C" int aid_adam(const aid_adam_params* p, void* stream) {
    AID_REQUIRE(p && p->param && p->grad && p->m && p->v && p->n > 0, "aid_adam: null pointer");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// dst = dst * rate + src * (1 - rate)      (trainer.py:288-304)
__global__ __launch_bounds__(256) void ema_kernel(const aid_ema_params p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    p.dst[i] = p.dst[i] * p.rate + p.src[i] * (1.f - p.rate);
}

extern "C" int aid_ema(const aid_ema_params* p, void* stream) {
    AID_REQUIRE(p && p->dst && p->src && p->n > 0, "aid_ema: null pointer");
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// partial sums of squares (fp64) in nblk blocks, then one block folds them into out[0] = sum, out[1] = clip coefficient
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const aid_sumsq_params p, int nblk) {
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < p.n; i += (int64_t)nblk * 256) { const double v = p.x[i]; s += v * v; }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) p.ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void sumsq_final_kernel(const aid_sumsq_params p, int nblk) {
    if (threadIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += p.ws[i];
    const float nrm = (float)sqrt(s);
    p.out[0] = nrm;
    p.out[1] = (p.max_norm > 0.f) ? fminf(1.f, p.max_norm / (nrm + 1e-6f)) : 1.f;   // torch.nn.utils.clip_grad_norm_
}

extern "C" int aid_sumsq(const aid_sumsq_params* p, void* stream) {
    AID_REQUIRE(p && p->x && p->ws && p->out && p->n > 0, "aid_sumsq: null pointer");
    const int nblk = AID_SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *p, nblk);
    AID_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *p, nblk);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// ------------------------------------------------------------------------------------------------------
// aid_pack_conv_weight: one launch writes every kernel-side layout of one conv weight [Cout,Cin,KH,KW]: the tap-major pack
// wp[KH*KW][Cin_pad][Cout_pad], its input-gradient operator (taps flipped, channel roles swapped) wpT, and for 5x3 layers the F(4,3) packs
// U = G w (fp64 arithmetic, rounded once) of both.  Replaces ~30 torch launches per layer after every optimiser step.
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const aid_pack_conv_weight_params p) {
    const int K = p.KH * p.KW;
    const int64_t n0 = (int64_t)K * p.Cin_pad * p.Cout_pad, n1 = p.wpT ? (int64_t)K * p.Cin_padT * p.Cout_padT : 0;
    const int64_t n2 = p.wpw ? (int64_t)30 * p.Cin_pad * p.Cout_pad : 0, n3 = p.wpwT ? (int64_t)30 * p.Cin_padT * p.Cout_padT : 0;
    const int64_t n4 = p.wpw8 ? (int64_t)50 * p.Cin_pad * p.Cout_pad : 0, n5 = p.wpw8T ? (int64_t)50 * p.Cin_padT * p.Cout_padT : 0;
    const int64_t n6 = p.wpw2 ? (int64_t)48 * p.Cin_pad * p.Cout_pad : 0, n7 = p.wpw2T ? (int64_t)48 * p.Cin_padT * p.Cout_padT : 0;
    const int64_t n8 = p.wpw3 ? (int64_t)80 * p.Cin_pad * p.Cout_pad : 0, n9 = p.wpw3T ? (int64_t)80 * p.Cin_padT * p.Cout_padT : 0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n0 + n1 + n2 + n3 + n4 + n5 + n6 + n7 + n8 + n9) return;
    int mode = 0;
    if (i >= n0) { i -= n0; mode = 1; if (i >= n1) { i -= n1; mode = 2; if (i >= n2) { i -= n2; mode = 3; if (i >= n3) { i -= n3; mode = 4; if (i >= n4) { i -= n4; mode = 5;
        if (i >= n5) { i -= n5; mode = 6; if (i >= n6) { i -= n6; mode = 7; if (i >= n7) { i -= n7; mode = 8; if (i >= n8) { i -= n8; mode = 9; } } } } } } } } }
    const bool tr = mode & 1;
    const int cip = tr ? p.Cin_padT : p.Cin_pad, cop = tr ? p.Cout_padT : p.Cout_pad;
    const int co = (int)(i % cop);                          // output channel of THIS operator (transposed: the layer's input channel)
    const int ci = (int)((i / cop) % cip);
    const int t = (int)(i / ((int64_t)cop * cip));
    const int nci = tr ? p.Cout : p.Cin, nco = tr ? p.Cin : p.Cout;
    float v = 0.f;
    if (ci < nci && co < nco) {
        auto W = [&](int kh, int kw) {                      // operator weight [co][ci][kh][kw]; transposed: w[ci][co][KH-1-kh][KW-1-kw]
            const int o = tr ? ci : co, c = tr ? co : ci;
            const int a = tr ? p.KH - 1 - kh : kh, b = tr ? p.KW - 1 - kw : kw;
            return p.w[(((int64_t)o * p.Cin + c) * p.KH + a) * p.KW + b];
        };
        if (mode < 2) {
            v = W(t / p.KW, t % p.KW);
        } else if (mode >= 8) {                             // 2-D form with F(8,3) along T: plane t = xf * 10 + xt, U = GF w G8^T (aid_wino45.h rows, aid_wino8.h samples)
            constexpr double GF[8][5] = AID_W45_GF;
            constexpr double G8[10][3] = AID_W8_G;
            const int xf = t / 10, xt = t - 10 * xf;
            double u = 0.0;
            for (int kh = 0; kh < 5; ++kh)
                u += GF[xf][kh] * (G8[xt][0] * (double)W(kh, 0) + G8[xt][1] * (double)W(kh, 1) + G8[xt][2] * (double)W(kh, 2));
            v = (float)u;
        } else if (mode >= 6) {                             // 2-D form: plane t = xf * 6 + xt, U = GF w GT^T (aid_wino45.h)
            constexpr double GF[8][5] = AID_W45_GF;
            constexpr double GT[6][3] = AID_W45_GT;
            const int xf = t / 6, xt = t - 6 * xf;
            double u = 0.0;
            for (int kh = 0; kh < 5; ++kh)
                u += GF[xf][kh] * (GT[xt][0] * (double)W(kh, 0) + GT[xt][1] * (double)W(kh, 1) + GT[xt][2] * (double)W(kh, 2));
            v = (float)u;
        } else if (mode >= 4) {                             // F(8,3): tap = xi * 5 + kh, U = G w (aid_wino8.h)
            constexpr double G8[10][3] = AID_W8_G;
            const int xi = t / 5, kh = t - 5 * xi;
            v = (float)(G8[xi][0] * (double)W(kh, 0) + G8[xi][1] * (double)W(kh, 1) + G8[xi][2] * (double)W(kh, 2));
        } else {                                            // tap = xi * 5 + kh
            const int xi = t / 5, kh = t - 5 * xi;
            const double w0 = (double)W(kh, 0), w1 = (double)W(kh, 1), w2 = (double)W(kh, 2);
            double u;
            switch (xi) {
                case 0: u = w0 / 4; break;
                case 1: u = -(w0 + w1 + w2) / 6; break;
                case 2: u = -(w0 - w1 + w2) / 6; break;
                case 3: u = (w0 + 2 * w1 + 4 * w2) / 24; break;
                case 4: u = (w0 - 2 * w1 + 4 * w2) / 24; break;
                default: u = w2;
            }
            v = (float)u;
        }
    }
    float* out = mode == 0 ? p.wp : (mode == 1 ? p.wpT : (mode == 2 ? p.wpw : (mode == 3 ? p.wpwT : (mode == 4 ? p.wpw8 : (mode == 5 ? p.wpw8T : (mode == 6 ? p.wpw2 : (mode == 7 ? p.wpw2T : (mode == 8 ? p.wpw3 : p.wpw3T))))))));
    out[i] = v;
}

extern "C" int aid_pack_conv_weight(const aid_pack_conv_weight_params* p, void* stream) {
    AID_REQUIRE(p && p->w && p->wp, "aid_pack_conv_weight: null pointer");
    AID_REQUIRE(p->Cin > 0 && p->Cout > 0 && p->KH >= 1 && p->KW >= 1, "aid_pack_conv_weight: bad shape");
    int cip, cop, cipT, copT;
    aid_conv2d_pack_dims(p->Cin, p->Cout, &cip, &cop);
    aid_conv2d_pack_dims(p->Cout, p->Cin, &cipT, &copT);
    AID_REQUIRE(p->Cin_pad == cip && p->Cout_pad == cop && (!p->wpT || (p->Cin_padT == cipT && p->Cout_padT == copT)), "aid_pack_conv_weight: padded dims != aid_conv2d_pack_dims()");
    AID_REQUIRE((!p->wpw && !p->wpwT && !p->wpw8 && !p->wpw8T && !p->wpw2 && !p->wpw2T && !p->wpw3 && !p->wpw3T) || (p->KH == 5 && p->KW == 3), "aid_pack_conv_weight: the Winograd packs are for 5x3 layers");
    AID_REQUIRE((!p->wpwT && !p->wpw8T && !p->wpw2T && !p->wpw3T) || p->wpT, "aid_pack_conv_weight: wpwT / wpw8T / wpw2T / wpw3T need wpT's dims");
    const int K = p->KH * p->KW;
    const int64_t n = (int64_t)K * cip * cop + (p->wpT ? (int64_t)K * cipT * copT : 0) + (p->wpw ? (int64_t)30 * cip * cop : 0) + (p->wpwT ? (int64_t)30 * cipT * copT : 0)
                    + (p->wpw8 ? (int64_t)50 * cip * cop : 0) + (p->wpw8T ? (int64_t)50 * cipT * copT : 0)
                    + (p->wpw2 ? (int64_t)48 * cip * cop : 0) + (p->wpw2T ? (int64_t)48 * cipT * copT : 0)
                    + (p->wpw3 ? (int64_t)80 * cip * cop : 0) + (p->wpw3T ? (int64_t)80 * cipT * copT : 0);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
