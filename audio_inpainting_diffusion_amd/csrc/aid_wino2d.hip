// The dilated 5x3 convolution of the C >= 128 layers in the NON-FUSED 2-D Winograd form F(4,5) x F(4,3) (round 5).
//
// The 1-D forms of aid_conv_wino.hip (F(4,3) / F(8,3) along T, the five dilated rows as five K steps) are bound by the fp32 matrix pipe:
// 7.5 / 6.25 MFMA products per output.  Transforming the dilated ROW axis as well -- F(4,5) on the sub-lattice of one dilation residue class --
// leaves 48 products per 4 x 4 outputs = 3.0 per output.  Fused into one kernel that does not map onto this machine (48 accumulator planes per
// tile); as three passes it does:
//   1. aid_scale_act(wino = 3) / the reverse sweep's gate pass: V[xi][ci][n] = BF^T gelu(x scale) BT, one HBM pass, 3x the activation out
//   2. w2d_gemm_kernel: 48 independent fp32-MFMA GEMMs  M_xi[co][n] = sum_ci U_xi[ci][co] V_xi[ci][n]   (U = GF w GT^T packed once)
//   3. w2d_output_kernel: y = AF^T M AT + the epilogue of the fused kernels (gate, residual, 1/sqrt2, dGELU, statistics / dot partials)
// Position index n = b * NB + (j * dil + r) * (T/4) + g:  4 x 4 output block of rows r + dil (4j .. 4j+3), samples 4g .. 4g+3 of sample b.
// Error (tools/wino2d_fm5_error.py, profiles/r04_wino2d_fm5_error.txt): 2.2e-6 .. 4.3e-6 rel-L2 per layer, below the shipped F(8,3) kernel.
// Matrices: tools/gen_wino45.py -> aid_wino45.h.
//
// Round 6: the T axis may also take F(8,3) (x_wino = 4; the ten-point transform of aid_wino8.h, the one the fused F(8,3) kernel uses): F(4,5) x F(8,3) has
// 8 x 10 = 80 products per 4 x 8 outputs = 2.5 per output (-17 % MFMAs) and its Winograd-domain tensors are 80 / 32 = 2.5 x the activation instead of 3 x
// (-17 % of the GEMM's bytes, -12 % / -9 % of the passes').  Same three kernels, templated on the T form TF = 4 | 8: NTP = TF + 2 planes along T,
// groups of TF samples (TG = T / TF), xi = xf * NTP + xt, position index as above with T/TF groups per row.  fp32 error 1.0e-5 (K = 128) / 1.3e-5 (K = 256)
// per layer (tools/wino2d_fm5_error.py --tf 8, profiles/r06_wino2d_f45x83_error.txt) against 2.8 / 4.3e-6 for F(4,5) x F(4,3).
#include "aid_common.h"
#include "aid_wino45.h"
#include "aid_wino8.h"
#include "aid_fin.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// direct-to-LDS load, scalar 64-bit base + 32-bit per-lane byte offset (see aid_conv_wino.hip: issued through inline asm so that the kernel,
// not the compiler's waitcnt pass, decides how long the loads stay in flight)
#define W2D_LDS_ADDR(lptr) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(lptr))
#define W2D_DMA16_SBASE(voff, sbase, lds_addr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory")
#define W2D_VMCNT(n) ((((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)))
#define W2D_LGKMCNT0 (15 | (7 << 4) | (0 << 8) | (3 << 14))

template <typename F, int... I>
__device__ __forceinline__ void w2d_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void w2d_static_for(F&& f) { w2d_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// =====================================================================================================================================
// 2. the batched GEMM.  Per transform index xi:  C[m][n] = sum_k A[k][m] B[k][n],  A = U_xi [Cin][Cout_pad], B = V_xi [Cin][N], C = M_xi [Cout][N].
// A workgroup computes M_BLK x N_BLK of one xi; both operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (KC channels per chunk, NBUF buffers,
// the loads of chunk c + NBUF - 1 issued one piece per k-step while chunk c is multiplied); LDS images are [k][m] / [k][n] exactly as in memory,
// a lane's MT (NT) fragments are INTERLEAVED rows (columns) m = MT*i + .. so that one ds_read_b64 / b128 feeds all of them and the epilogue
// stores float4 along n.  Two workgroups per CU: one tile's 128-KB store epilogue runs under the other's K loop.
// =====================================================================================================================================
struct W2dGemmDev {
    const float* U; const float* V; float* Mo;
    int Cin, Cout, Cin_pad, Cout_pad, N;
    int nchunks, ntn, ntm, ntiles, per_xcd;
};

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF>
struct W2dGemmShape {
    static constexpr int NW = WGM * WGN;
    static constexpr int M_BLK = 32 * MT * WGM, N_BLK = 32 * NT * WGN;
    static constexpr int XSZ = KC * N_BLK, WSZ = KC * M_BLK, BUFSZ = XSZ + WSZ;
    static constexpr int NXP = XSZ / 256, NWP = WSZ / 256, NP = NXP + NWP;
    static constexpr int PPW = (NP + NW - 1) / NW;
    static constexpr int NSTEP = KC / 2;
    static constexpr int LDS = NBUF * BUFSZ;
    static_assert(XSZ % 256 == 0 && WSZ % 256 == 0 && NP % NW == 0, "whole 1-KiB pieces, the same number for every wave");
};

template <int N> struct W2dVec;
template <> struct W2dVec<1> { typedef float type; };
template <> struct W2dVec<2> { typedef float2 type; };
template <> struct W2dVec<4> { typedef float4 type; };
struct w2d_float3 { float x, y, z; };                                  // (three interleaved rows of the 96-channel levels: 12-byte aligned, read as b32 / b64 pieces)
template <> struct W2dVec<3> { typedef w2d_float3 type; };
__device__ __forceinline__ float w2d_get(const w2d_float3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ float w2d_get(const float& v, int) { return v; }
__device__ __forceinline__ float w2d_get(const float2& v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float w2d_get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

template <int NMAX>
__device__ __forceinline__ void w2d_wait_vmcnt(int n) {
    if constexpr (NMAX == 0) { __builtin_amdgcn_s_waitcnt(W2D_VMCNT(0)); }
    else { if (n >= NMAX) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(NMAX)); else w2d_wait_vmcnt<NMAX - 1>(n); }
}

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF, int WPC>
__global__ __launch_bounds__(64 * WGM * WGN, (WPC * WGM * WGN + 3) / 4) void w2d_gemm_kernel(const W2dGemmDev a) {
    using S = W2dGemmShape<MT, NT, WGM, WGN, KC, NBUF>;
    constexpr int NW = S::NW, M_BLK = S::M_BLK, N_BLK = S::N_BLK, XSZ = S::XSZ, BUFSZ = S::BUFSZ;
    constexpr int NXP = S::NXP, PPW = S::PPW, NSTEP = S::NSTEP;
    constexpr int ISTEPS = NSTEP < PPW ? NSTEP : PPW;          // k-steps of a chunk in which the pieces of a later chunk are issued
    typedef typename W2dVec<MT>::type avec;
    typedef typename W2dVec<NT>::type bvec;
    __shared__ __attribute__((aligned(16))) float smem[S::LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int half = lane >> 5;

    // XCD-aware tile order (hardware deals consecutive workgroup ids round-robin to the 8 XCDs): every XCD walks ONE contiguous range of
    // (xi, n tile, m tile) triples, m tile fastest -- the Cout tiles of the same V columns meet in one L2, and so do the tiles of one xi (its U slice).
    const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (Lt >= a.ntiles) return;
    int rest = Lt;
    const int by = rest % a.ntm; rest /= a.ntm;
    const int bn = rest % a.ntn;
    const int xi = rest / a.ntn;
    const int m0 = by * M_BLK, n0 = bn * N_BLK;

    // A piece = 1 KiB of LDS filled by one wave instruction (16 bytes per lane); per piece and lane ONE 32-bit byte offset from a wave-uniform base
    // that advances by a scalar stride per chunk.  Columns past N (the last n tile) are clamped onto the last four real columns: every lane always
    // loads (the per-chunk instruction count the vmcnt waits rely on is exact) and the duplicate columns are never stored.
    unsigned poff[PPW];
    int plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        plds[i] = pc * 256;
        if (pc < NXP) {
            const int e = pc * 256 + 4 * lane;
            const int k = e / N_BLK, nl = e % N_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.N + min(n0 + nl, a.N - 4)));
        } else {
            const int e = (pc - NXP) * 256 + 4 * lane;
            const int k = e / M_BLK, col = e % M_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.Cout_pad + m0 + col));
        }
    }
    const int64_t xstep = (int64_t)KC * a.N * 4, wstep = (int64_t)KC * a.Cout_pad * 4;          // bytes per chunk (wave-uniform)
    const char* const xbase = reinterpret_cast<const char*>(a.V + (int64_t)xi * a.Cin * a.N);
    const char* const wbase = reinterpret_cast<const char*>(a.U + (int64_t)xi * a.Cin_pad * a.Cout_pad);

    const int vB = half * N_BLK + wn * (32 * NT) + NT * (lane & 31);
    const int vA = XSZ + half * M_BLK + wm * (32 * MT) + MT * (lane & 31);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_piece = [&](auto ic, int ch, float* buf) {
        constexpr int i = decltype(ic)::value;
        const char* base = (wave + i * NW < NXP) ? xbase + ch * xstep : wbase + ch * wstep;          // (scalar)
        const unsigned la = W2D_LDS_ADDR(buf + plds[i]);
        const unsigned off = poff[i];
        W2D_DMA16_SBASE(off, base, la);
    };
    auto issue_all = [&](int ch, float* buf) { w2d_static_for<PPW>([&](auto ic) { issue_piece(ic, ch, buf); }); };
    auto issue_step = [&](auto sc, int ch, float* buf) {
        w2d_static_for<PPW>([&](auto ic) {
            if constexpr (decltype(sc)::value < ISTEPS && decltype(ic)::value % ISTEPS == decltype(sc)::value) issue_piece(ic, ch, buf);
        });
    };
    const int nch = a.nchunks;
    constexpr int AHEAD = NBUF - 1;
    w2d_static_for<AHEAD>([&](auto qc) { if (decltype(qc)::value < nch) issue_all(decltype(qc)::value, smem + decltype(qc)::value * BUFSZ); });
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = smem + cur * BUFSZ;
        float* Nx = smem + ((cur + AHEAD) % NBUF) * BUFSZ;      // read during the previous chunk (every wave is past that chunk's barrier)
        const bool more = (ch + AHEAD) < nch;
        bvec bv[2];
        avec av[2];
        auto load_step = [&](int s, int q) {
            bv[q] = *reinterpret_cast<const bvec*>(Bf + vB + 2 * s * N_BLK);
            av[q] = *reinterpret_cast<const avec*>(Bf + vA + 2 * s * M_BLK);
        };
        load_step(0, 0);
        w2d_static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (more) issue_step(sc, ch + AHEAD, Nx);
            if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
            constexpr int q = s & 1;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2d_get(av[q], i), w2d_get(bv[q], j), acc[i][j], 0, 0, 0);
        });
        asm volatile("" ::: "memory");
        // the loads of chunk ch + 1 must have landed; with three or more buffers those issued during this chunk (PPW per wave) may stay in flight
        if (NBUF >= 3 && more) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 1)));
        else if (NBUF >= 4 && (ch + AHEAD - 1) < nch) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 2)));
        else __builtin_amdgcn_s_waitcnt(W2D_VMCNT(0));
        __builtin_amdgcn_s_waitcnt(W2D_LGKMCNT0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int ch = 0; ch < nch; ch += NBUF)
        w2d_static_for<NBUF>([&](auto qc) { if (ch + decltype(qc)::value < nch) chunk(qc, ch + decltype(qc)::value); });

    // ---- epilogue: NT consecutive columns per lane and row ------------------------------------------------------------------------------
    const int n = n0 + wn * (32 * NT) + NT * (lane & 31);
    if (n >= a.N) return;
    float* const cb = a.Mo + (int64_t)xi * a.Cout * a.N + n;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (32 * MT) + MT * (4 * half + (r & 3) + 8 * (r >> 2)) + i;
            if (m >= a.Cout) continue;
            float* q = cb + (int64_t)m * a.N;
            if constexpr (NT == 4) *reinterpret_cast<float4*>(q) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            else if constexpr (NT == 2) *reinterpret_cast<float2*>(q) = make_float2(acc[i][0][r], acc[i][1][r]);
            else *q = acc[i][0][r];
        }
}

// ---- LABELLED VARIANT (never the default, never the headline): the same batched GEMM with both fp32 operands SPLIT into three bf16
// pieces each (truncation: 8 + 8 + 8 significand bits, x = p0 + p1 + p2 EXACTLY) and the six largest cross products on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16: a bf16 x bf16 product is exact in fp32, accumulation in fp32) -- p2 q0 + p0 q2 + p1 q1 + p1 q0 + p0 q1 + p0 q0, smallest first;
// the dropped terms p1 q2 + p2 q1 (+ p2 q2) are NOT negligible per product: each is ~2^-24 of it (measured, tools/bf16split_error.py --per-product,
// profiles/r06_bf16split_per_product.txt: max 2^-21.3, median 2^-25.2 -- up to 6x one fp32 rounding; only p2 q2 is below 2^-32).  Error study BEFORE the kernel
// (tools/bf16split_error.py, profiles/r05_bf16split_error.txt): 1.8e-6 per layer at K = 256 against 4.3e-6 of the fp32-MFMA form -- that comes from the bf16 MFMA
// accumulating 16 exact products per instruction (fewer fp32 roundings of the running sum), not from exact products; the 3-product split is 6e-5: not built.
// 6 MFMAs of 32 cycles per 16 k against 8 of 64: 0.375x the matrix-pipe time; the split is VALU work on the fragments (4 ops + 1.5 packs per value).
// Same LDS images, same direct-to-LDS staging, one k-step of 16 per chunk: lane half h holds k = 8 h .. 8 h + 7 of BOTH fragments.
typedef __bf16 w2d_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w2d_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void w2d_split3(const float (&x)[8], w2d_u32x4 (&p)[3]) {
    unsigned a[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned b0 = __float_as_uint(x[k]) & 0xffff0000u;
        const float r1 = x[k] - __uint_as_float(b0);                 // exact
        const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(b1);                   // exact, at most 8 significant bits: its bf16 truncation is itself
        a[0][k] = b0; a[1][k] = b1; a[2][k] = __float_as_uint(r2);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) p[q][k] = __builtin_amdgcn_perm(a[q][2 * k + 1], a[q][2 * k], 0x07060302u);     // (hi16 of k+1) << 16 | hi16 of k
}

template <int MT, int NT, int WGM, int WGN, int NBUF, int WPC>
__global__ __launch_bounds__(64 * WGM * WGN, (WPC * WGM * WGN + 3) / 4) void w2d_gemm_s6_kernel(const W2dGemmDev a) {
    constexpr int KC = 16;
    using S = W2dGemmShape<MT, NT, WGM, WGN, KC, NBUF>;
    constexpr int NW = S::NW, M_BLK = S::M_BLK, N_BLK = S::N_BLK, XSZ = S::XSZ, BUFSZ = S::BUFSZ;
    constexpr int NXP = S::NXP, PPW = S::PPW;
    typedef typename W2dVec<MT>::type avec;
    typedef typename W2dVec<NT>::type bvec;
    __shared__ __attribute__((aligned(16))) float smem[S::LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int half = lane >> 5;
    const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (Lt >= a.ntiles) return;
    int rest = Lt;
    const int by = rest % a.ntm; rest /= a.ntm;
    const int bn = rest % a.ntn;
    const int xi = rest / a.ntn;
    const int m0 = by * M_BLK, n0 = bn * N_BLK;

    unsigned poff[PPW];
    int plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        plds[i] = pc * 256;
        if (pc < NXP) {
            const int e = pc * 256 + 4 * lane;
            const int k = e / N_BLK, nl = e % N_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.N + min(n0 + nl, a.N - 4)));
        } else {
            const int e = (pc - NXP) * 256 + 4 * lane;
            const int k = e / M_BLK, col = e % M_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.Cout_pad + m0 + col));
        }
    }
    const int64_t xstep = (int64_t)KC * a.N * 4, wstep = (int64_t)KC * a.Cout_pad * 4;
    const char* const xbase = reinterpret_cast<const char*>(a.V + (int64_t)xi * a.Cin * a.N);
    const char* const wbase = reinterpret_cast<const char*>(a.U + (int64_t)xi * a.Cin_pad * a.Cout_pad);
    const int vB = (8 * half) * N_BLK + wn * (32 * NT) + NT * (lane & 31);
    const int vA = XSZ + (8 * half) * M_BLK + wm * (32 * MT) + MT * (lane & 31);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_all = [&](int ch, float* buf) {
        w2d_static_for<PPW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const char* base = (wave + i * NW < NXP) ? xbase + ch * xstep : wbase + ch * wstep;
            const unsigned la = W2D_LDS_ADDR(buf + plds[i]);
            const unsigned off = poff[i];
            W2D_DMA16_SBASE(off, base, la);
        });
    };
    const int nch = a.nchunks;
    constexpr int AHEAD = NBUF - 1;
    w2d_static_for<AHEAD>([&](auto qc) { if (decltype(qc)::value < nch) issue_all(decltype(qc)::value, smem + decltype(qc)::value * BUFSZ); });
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = smem + cur * BUFSZ;
        float* Nx = smem + ((cur + AHEAD) % NBUF) * BUFSZ;
        const bool more = (ch + AHEAD) < nch;
        if (more) issue_all(ch + AHEAD, Nx);
        avec av[8];
        bvec bv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            av[k] = *reinterpret_cast<const avec*>(Bf + vA + k * M_BLK);
            bv[k] = *reinterpret_cast<const bvec*>(Bf + vB + k * N_BLK);
        }
        w2d_u32x4 pa[MT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = w2d_get(av[k], i);
            w2d_split3(x, pa[i]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = w2d_get(bv[k], j);
            w2d_u32x4 pb[3];
            w2d_split3(x, pb);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x16 c = acc[i][j];
#define W2D_S6(qa, qb) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w2d_bf16x8, pa[i][qa]), __builtin_bit_cast(w2d_bf16x8, pb[qb]), c, 0, 0, 0)
                W2D_S6(2, 0); W2D_S6(0, 2); W2D_S6(1, 1); W2D_S6(1, 0); W2D_S6(0, 1); W2D_S6(0, 0);
#undef W2D_S6
                acc[i][j] = c;
            }
        }
        asm volatile("" ::: "memory");
        if (NBUF >= 3 && more) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 1)));
        else if (NBUF >= 4 && (ch + AHEAD - 1) < nch) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 2)));
        else __builtin_amdgcn_s_waitcnt(W2D_VMCNT(0));
        __builtin_amdgcn_s_waitcnt(W2D_LGKMCNT0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int ch = 0; ch < nch; ch += NBUF)
        w2d_static_for<NBUF>([&](auto qc) { if (ch + decltype(qc)::value < nch) chunk(qc, ch + decltype(qc)::value); });

    const int n = n0 + wn * (32 * NT) + NT * (lane & 31);
    if (n >= a.N) return;
    float* const cb = a.Mo + (int64_t)xi * a.Cout * a.N + n;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (32 * MT) + MT * (4 * half + (r & 3) + 8 * (r >> 2)) + i;
            if (m >= a.Cout) continue;
            float* q = cb + (int64_t)m * a.N;
            if constexpr (NT == 4) *reinterpret_cast<float4*>(q) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            else if constexpr (NT == 2) *reinterpret_cast<float2*>(q) = make_float2(acc[i][0][r], acc[i][1][r]);
            else *q = acc[i][0][r];
        }
}

template <int MT, int NT, int WGM, int WGN, int NBUF, int WPC>
static int w2d_launch_gemm_s6(const aid_wino2d_gemm_params* p, hipStream_t st, const char* name) {
    using S = W2dGemmShape<MT, NT, WGM, WGN, 16, NBUF>;
    AID_REQUIRE(p->Cin % 16 == 0, "aid_wino2d_gemm: Cin must be a multiple of the K chunk");
    AID_REQUIRE(p->Cout_pad % S::M_BLK == 0, "aid_wino2d_gemm: Cout_pad must be a multiple of the M tile");
    AID_REQUIRE((int64_t)p->Cin * p->N * 4 < (1LL << 32) && (int64_t)p->Cin_pad * p->Cout_pad * 4 < (1LL << 32), "aid_wino2d_gemm: one plane must stay below 4 GiB");
    W2dGemmDev a;
    a.U = p->U; a.V = p->V; a.Mo = p->M;
    a.Cin = p->Cin; a.Cout = p->Cout; a.Cin_pad = p->Cin_pad; a.Cout_pad = p->Cout_pad; a.N = (int)p->N;
    a.nchunks = p->Cin / 16;
    a.ntn = aid_cdiv(p->N, S::N_BLK);
    a.ntm = aid_cdiv(p->Cout, S::M_BLK);
    a.ntiles = p->nxi * a.ntn * a.ntm;
    a.per_xcd = aid_cdiv(a.ntiles, 8);
    hipLaunchKernelGGL((w2d_gemm_s6_kernel<MT, NT, WGM, WGN, NBUF, WPC>), dim3((unsigned)(8 * a.per_xcd)), dim3(64 * S::NW), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel(name);
    return AID_OK;
}

// =====================================================================================================================================
// 2b. the batched GEMM with the ROW-AXIS OUTPUT TRANSFORM FOLDED IN (round 6; the HBM-bound launches: Cin <= 128 on the 80-plane form).
// "halve M".  A workgroup owns one T-axis index xt and one (Cout tile, column tile) and walks the EIGHT row-axis planes
// xf = 0 .. 7 of it one after the other through the same three-buffer direct-to-LDS pipeline (the chunk stream is continuous: chunk c is K-chunk
// c % nch of plane c / nch); when a plane's K loop ends its accumulators are folded into the four output-row accumulators with the column of AF^T
// (Y[j] += AF^T[j][xf] * acc) and cleared.  It writes M'[j * NTP + xt][co][n], 4 NTP planes instead of 8 NTP: the GEMM's writes and the output
// pass's reads of M halve (these launches sit below the fp32 ridge at 24-32 FLOP/B), the output pass keeps only its T transform.
// Five accumulator sets per tile: 64-column tiles (one 32 x 32 fragment column pair per wave), two workgroups per CU.
// =====================================================================================================================================
__constant__ float w2d_atf[8][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f}, {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f},
                                    {1.f, 0.5f, 0.25f, 0.125f}, {1.f, -0.5f, 0.25f, -0.125f}, {0.f, 0.f, 0.f, 1.f}};      // [xf][j] = AF^T[j][xf] (aid_wino45.h)

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF, int WPC>
__global__ __launch_bounds__(64 * WGM * WGN, (WPC * WGM * WGN + 3) / 4) void w2d_gemm_fold_kernel(const W2dGemmDev a, const int ntp) {
    using S = W2dGemmShape<MT, NT, WGM, WGN, KC, NBUF>;
    constexpr int NW = S::NW, M_BLK = S::M_BLK, N_BLK = S::N_BLK, XSZ = S::XSZ, BUFSZ = S::BUFSZ;
    constexpr int NXP = S::NXP, PPW = S::PPW, NSTEP = S::NSTEP;
    constexpr int ISTEPS = NSTEP < PPW ? NSTEP : PPW;
    typedef typename W2dVec<MT>::type avec;
    typedef typename W2dVec<NT>::type bvec;
    __shared__ __attribute__((aligned(16))) float smem[S::LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int half = lane >> 5;

    const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);      // XCD-aware order: (xt, n tile, m tile), m fastest
    if (Lt >= a.ntiles) return;
    int rest = Lt;
    const int by = rest % a.ntm; rest /= a.ntm;
    const int bn = rest % a.ntn;
    const int xt = rest / a.ntn;
    const int m0 = by * M_BLK, n0 = bn * N_BLK;

    unsigned poff[PPW];
    int plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        plds[i] = pc * 256;
        if (pc < NXP) {
            const int e = pc * 256 + 4 * lane;
            const int k = e / N_BLK, nl = e % N_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.N + min(n0 + nl, a.N - 4)));
        } else {
            const int e = (pc - NXP) * 256 + 4 * lane;
            const int k = e / M_BLK, col = e % M_BLK;
            poff[i] = (unsigned)(4 * ((int64_t)k * a.Cout_pad + min(m0 + col, a.Cout_pad - 4)));      // a 96-channel panel on the 128-wide tile: columns past Cout_pad re-read the last four (never stored)
        }
    }
    const int nch = a.nchunks, nct = 8 * nch;                // chunks per plane, chunks of the whole stream
    const int64_t xstep = (int64_t)KC * a.N * 4, wstep = (int64_t)KC * a.Cout_pad * 4;          // bytes per chunk
    const int64_t xplane = (int64_t)ntp * a.Cin * a.N * 4, wplane = (int64_t)ntp * a.Cin_pad * a.Cout_pad * 4;      // bytes from plane (xf, xt) to (xf + 1, xt)
    const char* const xbase = reinterpret_cast<const char*>(a.V + (int64_t)xt * a.Cin * a.N);
    const char* const wbase = reinterpret_cast<const char*>(a.U + (int64_t)xt * a.Cin_pad * a.Cout_pad);

    const int vB = half * N_BLK + wn * (32 * NT) + NT * (lane & 31);
    const int vA = XSZ + half * M_BLK + wm * (32 * MT) + MT * (lane & 31);

    f32x16 acc[MT][NT];
    f32x16 Y[4][MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; Y[0][i][j][r] = 0.f; Y[1][i][j][r] = 0.f; Y[2][i][j][r] = 0.f; Y[3][i][j][r] = 0.f; }

    auto issue_piece = [&](auto ic, const char* xs, const char* ws, float* buf) {
        constexpr int i = decltype(ic)::value;
        const char* base = (wave + i * NW < NXP) ? xs : ws;          // (scalar)
        const unsigned la = W2D_LDS_ADDR(buf + plds[i]);
        const unsigned off = poff[i];
        W2D_DMA16_SBASE(off, base, la);
    };
    auto src_of = [&](int c, const char*& xs, const char*& ws) {      // (scalar) sources of stream chunk c
        const int xf = c / nch, kc = c - xf * nch;
        xs = xbase + xf * xplane + kc * xstep;
        ws = wbase + xf * wplane + kc * wstep;
    };
    constexpr int AHEAD = NBUF - 1;
    w2d_static_for<AHEAD>([&](auto qc) {
        const char *xs, *ws;
        src_of(decltype(qc)::value, xs, ws);
        w2d_static_for<PPW>([&](auto ic) { issue_piece(ic, xs, ws, smem + decltype(qc)::value * BUFSZ); });
    });
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = smem + cur * BUFSZ;
        float* Nx = smem + ((cur + AHEAD) % NBUF) * BUFSZ;
        const bool more = (ch + AHEAD) < nct;
        const char *xs = xbase, *ws = wbase;
        if (more) src_of(ch + AHEAD, xs, ws);
        bvec bv[2];
        avec av[2];
        auto load_step = [&](int s_, int q) {
            bv[q] = *reinterpret_cast<const bvec*>(Bf + vB + 2 * s_ * N_BLK);
            av[q] = *reinterpret_cast<const avec*>(Bf + vA + 2 * s_ * M_BLK);
        };
        load_step(0, 0);
        w2d_static_for<NSTEP>([&](auto sc) {
            constexpr int s_ = decltype(sc)::value;
            if (more) w2d_static_for<PPW>([&](auto ic) {
                if constexpr (s_ < ISTEPS && decltype(ic)::value % ISTEPS == s_) issue_piece(ic, xs, ws, Nx);
            });
            if (s_ + 1 < NSTEP) load_step(s_ + 1, (s_ + 1) & 1);
            constexpr int q = s_ & 1;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2d_get(av[q], i), w2d_get(bv[q], j), acc[i][j], 0, 0, 0);
        });
        asm volatile("" ::: "memory");
        if (NBUF >= 3 && more) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 1)));
        else if (NBUF >= 4 && (ch + AHEAD - 1) < nct) __builtin_amdgcn_s_waitcnt(W2D_VMCNT(PPW * (AHEAD - 2)));
        else __builtin_amdgcn_s_waitcnt(W2D_VMCNT(0));
        __builtin_amdgcn_s_waitcnt(W2D_LGKMCNT0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int xf = ch / nch;
        if (ch - xf * nch == nch - 1) {                       // (wave-uniform) the K loop of plane xf is complete: fold it into the four output rows
            const float c0 = w2d_atf[xf][0], c1 = w2d_atf[xf][1], c2 = w2d_atf[xf][2], c3 = w2d_atf[xf][3];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        Y[0][i][j][r] += c0 * v; Y[1][i][j][r] += c1 * v; Y[2][i][j][r] += c2 * v; Y[3][i][j][r] += c3 * v;
                        acc[i][j][r] = 0.f;
                    }
        }
    };
    for (int ch = 0; ch < nct; ch += NBUF)
        w2d_static_for<NBUF>([&](auto qc) { if (ch + decltype(qc)::value < nct) chunk(qc, ch + decltype(qc)::value); });

    // ---- epilogue: the four output-row planes j * ntp + xt ---------------------------------------------------------------------------------
    const int n = n0 + wn * (32 * NT) + NT * (lane & 31);
    if (n >= a.N) return;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        float* const cb = a.Mo + (int64_t)(jr * ntp + xt) * a.Cout * a.N + n;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * MT) + MT * (4 * half + (r & 3) + 8 * (r >> 2)) + i;
                if (m >= a.Cout) continue;
                float* q = cb + (int64_t)m * a.N;
                if constexpr (NT == 4) *reinterpret_cast<float4*>(q) = make_float4(Y[jr][i][0][r], Y[jr][i][1][r], Y[jr][i][2][r], Y[jr][i][3][r]);
                else if constexpr (NT == 2) *reinterpret_cast<float2*>(q) = make_float2(Y[jr][i][0][r], Y[jr][i][1][r]);
                else *q = Y[jr][i][0][r];
            }
    }
}

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF, int WPC>
static int w2d_launch_gemm_fold(const aid_wino2d_gemm_params* p, int ntp, hipStream_t st, const char* name) {
    using S = W2dGemmShape<MT, NT, WGM, WGN, KC, NBUF>;
    AID_REQUIRE(p->Cin % KC == 0 && p->Cout_pad % 32 == 0 && p->nxi == 8 * ntp, "w2d gemm (folded): Cin % KC, Cout_pad % 32, nxi = 8 ntp");
    AID_REQUIRE((int64_t)p->Cin * p->N * 4 * p->nxi < (1LL << 40) && (int64_t)p->Cin * p->N * 4 < (1LL << 32), "w2d gemm (folded): plane size");
    W2dGemmDev a;
    a.U = p->U; a.V = p->V; a.Mo = p->M;
    a.Cin = p->Cin; a.Cout = p->Cout; a.Cin_pad = p->Cin_pad; a.Cout_pad = p->Cout_pad; a.N = (int)p->N;
    a.nchunks = p->Cin / KC;
    a.ntn = aid_cdiv(p->N, S::N_BLK);
    a.ntm = aid_cdiv(p->Cout, S::M_BLK);
    a.ntiles = ntp * a.ntn * a.ntm;
    a.per_xcd = aid_cdiv(a.ntiles, 8);
    hipLaunchKernelGGL((w2d_gemm_fold_kernel<MT, NT, WGM, WGN, KC, NBUF, WPC>), dim3((unsigned)(8 * a.per_xcd)), dim3(64 * S::NW), 0, st, a, ntp);
    AID_CHECK_LAUNCH();
    aid_note_kernel(name);
    return AID_OK;
}

// the split-precision variant as a process-wide switch of aid_conv2d(x_wino = 3) (0: the fp32-MFMA product kernel; 6: the six-product bf16 split): set
// by bench.py --mfma-split 6 and the variant's tests only
static int g_w2d_split = 0;
extern "C" int aid_wino2d_set_split(int pieces) {
    AID_REQUIRE(pieces == 0 || pieces == 6, "aid_wino2d_set_split: 0 (fp32 MFMA) or 6 (six bf16 products per fp32 product)");
    g_w2d_split = pieces;
    return AID_OK;
}

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF, int WPC>
static int w2d_launch_gemm(const aid_wino2d_gemm_params* p, hipStream_t st, const char* name, const char* name8 = nullptr) {
    if (!name8) name8 = name;
    using S = W2dGemmShape<MT, NT, WGM, WGN, KC, NBUF>;
    AID_REQUIRE(p->Cin % KC == 0, "aid_wino2d_gemm: Cin must be a multiple of the K chunk");
    AID_REQUIRE(p->Cout_pad % S::M_BLK == 0, "aid_wino2d_gemm: Cout_pad must be a multiple of the M tile");
    AID_REQUIRE((int64_t)p->Cin * p->N * 4 < (1LL << 32) && (int64_t)p->Cin_pad * p->Cout_pad * 4 < (1LL << 32), "aid_wino2d_gemm: one plane must stay below 4 GiB");
    W2dGemmDev a;
    a.U = p->U; a.V = p->V; a.Mo = p->M;
    a.Cin = p->Cin; a.Cout = p->Cout; a.Cin_pad = p->Cin_pad; a.Cout_pad = p->Cout_pad; a.N = (int)p->N;
    a.nchunks = p->Cin / KC;
    a.ntn = aid_cdiv(p->N, S::N_BLK);
    a.ntm = aid_cdiv(p->Cout, S::M_BLK);
    a.ntiles = p->nxi * a.ntn * a.ntm;
    a.per_xcd = aid_cdiv(a.ntiles, 8);
    hipLaunchKernelGGL((w2d_gemm_kernel<MT, NT, WGM, WGN, KC, NBUF, WPC>), dim3((unsigned)(8 * a.per_xcd)), dim3(64 * S::NW), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel(p->nxi == 80 ? name8 : name);          // (bench.py prices the launch by the form: 48 planes = 1/5, 80 planes = 1/6 of the direct-form FLOPs)
    return AID_OK;
}

extern "C" int aid_wino2d_gemm(const aid_wino2d_gemm_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->U && p->V && p->M, "aid_wino2d_gemm: null pointer");
    AID_REQUIRE(p->nxi > 0 && p->Cin > 0 && p->Cout > 0 && p->N > 0 && (p->N % 4) == 0 && p->N < (1LL << 31), "aid_wino2d_gemm: bad shape (N % 4 == 0)");
    AID_REQUIRE(((uintptr_t)p->U & 15) == 0 && ((uintptr_t)p->V & 15) == 0 && ((uintptr_t)p->M & 15) == 0 && (p->Cout_pad % 4) == 0, "aid_wino2d_gemm: 16-byte alignment");
    AID_REQUIRE(p->Cin_pad >= p->Cin && p->Cout_pad >= p->Cout, "aid_wino2d_gemm: padded dims");
    int variant = p->variant;
    if (variant == 0 && (p->Cout_pad % 128) != 0) {                    // the 96-channel levels: 96 x 128 tiles on two waves (three interleaved 32-row fragments per lane),
        AID_REQUIRE((p->Cout_pad % 96) == 0, "aid_wino2d_gemm: Cout_pad must be a multiple of 128 or of 96");      // three workgroups per CU (round 5's patch, kept out of the
        return w2d_launch_gemm<3, 2, 1, 2, 16, 3, 3>(p, st, "w2d_gemm_kernel<96x128,kc16,nb3,wpc3>", "w2d_gemm_kernel<96x128,kc16,nb3,wpc3>+t8");     // product then: with 48 planes it lost 0.6-1.2 %)
    }
    if (variant == 0 && g_w2d_split == 6) variant = ((int64_t)p->nxi * aid_cdiv(p->N, 256) * aid_cdiv(p->Cout, 128) < 1600) ? 101 : 100;
    if (variant == 0) {
        // 128 x 256 tiles (two workgroups per CU) while the launch has a few rounds of them; 128 x 128 tiles at four workgroups per CU for the short ones
        // (batch 1-2, the deepest level): 15-20 % faster there (profiles/r05_w2d_layer_probe2_b1.txt).  Every output element sums its K products in
        // the same order under either tiling: the choice does not change a bit of the result.
        const int64_t tiles = (int64_t)p->nxi * aid_cdiv(p->N, 256) * aid_cdiv(p->Cout, 128);
        variant = tiles < 1600 ? 11 : 12;
    }
    switch (variant) {
        case 12: return w2d_launch_gemm<2, 4, 2, 2, 16, 3, 2>(p, st, "w2d_gemm_kernel<128x256,kc16,nb3>", "w2d_gemm_kernel<128x256,kc16,nb3>+t8");
#ifdef AID_EXPERIMENT       // tile-shape probes of round 5 (tools/w2d_probe.py gemm, profiles/r05_w2d_gemm_probe1.txt): -DAID_EXPERIMENT builds only
        case 1: return w2d_launch_gemm<2, 4, 2, 2, 8, 3, 2>(p, st, "w2d_gemm_kernel<128x256,kc8,nb3>");
        case 2: return w2d_launch_gemm<2, 4, 2, 2, 16, 2, 2>(p, st, "w2d_gemm_kernel<128x256,kc16,nb2>");
        case 3: return w2d_launch_gemm<4, 2, 2, 2, 16, 3, 2>(p, st, "w2d_gemm_kernel<256x128,kc16,nb3>");
        case 4: return w2d_launch_gemm<2, 2, 2, 2, 16, 3, 3>(p, st, "w2d_gemm_kernel<128x128,kc16,nb3,wpc3>");
        case 5: return w2d_launch_gemm<2, 4, 2, 2, 8, 4, 2>(p, st, "w2d_gemm_kernel<128x256,kc8,nb4>");
        case 6: return w2d_launch_gemm<2, 2, 2, 4, 16, 3, 1>(p, st, "w2d_gemm_kernel<128x256,8waves,kc16,nb3>");
        case 7: return w2d_launch_gemm<2, 2, 2, 2, 8, 4, 3>(p, st, "w2d_gemm_kernel<128x128,kc8,nb4,wpc3>");
        case 8: return w2d_launch_gemm<2, 2, 2, 2, 8, 4, 4>(p, st, "w2d_gemm_kernel<128x128,kc8,nb4,wpc4>");
        case 9: return w2d_launch_gemm<2, 4, 2, 2, 8, 6, 2>(p, st, "w2d_gemm_kernel<128x256,kc8,nb6>");
        case 10: return w2d_launch_gemm<4, 2, 2, 2, 8, 4, 2>(p, st, "w2d_gemm_kernel<256x128,kc8,nb4>");
        case 102: return w2d_launch_gemm_s6<2, 4, 2, 2, 2, 2>(p, st, "w2d_gemm_s6_kernel<128x256,nb2>");
#endif
        case 11: return w2d_launch_gemm<2, 2, 2, 2, 16, 2, 4>(p, st, "w2d_gemm_kernel<128x128,kc16,nb2,wpc4>", "w2d_gemm_kernel<128x128,kc16,nb2,wpc4>+t8");
        case 100: return w2d_launch_gemm_s6<2, 4, 2, 2, 3, 2>(p, st, "w2d_gemm_s6_kernel<128x256,nb3>");
        case 101: return w2d_launch_gemm_s6<2, 2, 2, 2, 3, 3>(p, st, "w2d_gemm_s6_kernel<128x128,nb3,wpc3>");
        default: break;
    }
    aid_set_error("aid_wino2d_gemm: unknown variant (0 = automatic, 11 / 12 = the two product tilings, 100 / 101 = the labelled split-precision variant; "
                  "the round-5 probe tilings 1-10 / 102 exist in -DAID_EXPERIMENT builds only)");
    return AID_E_BADARG;
}

// =====================================================================================================================================
// Geometry shared by the three passes.  R = F / dil rows per residue class, J = ceil(R / 4) row tiles, TG = T / 4 sample groups,
// NB = J * dil * TG positions per sample, N = B * NB.  Position n' = (j * dil + r) * TG + g.
// =====================================================================================================================================
struct W2dGeo { int R, J, TG, NB; int64_t N; };
static inline W2dGeo w2d_geo(int B, int F, int T, int dil, int TF = 4) {
    W2dGeo g;
    g.R = F / dil; g.J = (g.R + 3) / 4; g.TG = T / TF; g.NB = g.J * dil * g.TG; g.N = (int64_t)B * g.NB;
    return g;
}
static inline int w2d_tf_of(int x_wino) { return x_wino == 4 ? 8 : 4; }          // aid_conv2d_params::x_wino / aid_scale_act_params::wino: 3 -> F(4,3), 4 -> F(8,3) along T
static inline bool w2d_shape_ok(int Cin, int Cout, int F, int T, int dil) {
    return dil >= 1 && (F % dil) == 0 && (T % 16) == 0 && T >= 16 && T <= 2048 && (Cin % 16) == 0 && ((Cout % 128) == 0 || (Cout % 96) == 0) && Cin >= 64;
}

// =====================================================================================================================================
// 1. input pass: V[xi][c][n] = BF^T act(x * scale[b,c]) BT.  A workgroup owns, for one (sample, channel), RB residue classes x JB row tiles x all
// sample groups.  Phase 1: the 4 JB + 4 sub-lattice rows of those classes (RB adjacent rows of T samples per sub-lattice index: contiguous in
// memory) are loaded as float4, scaled and activated ONCE per element and parked in LDS (36 KB: four workgroups per CU).  Phase 2: one tile per
// thread and turn -- 8 rows x 6 samples out of LDS, the T transform per row, the row transform per T plane, 48 stores.  Threads run along the
// position index n' = (j dil + r) TG + g, the memory order of every plane of V: a wave stores 256 contiguous bytes per plane (RB TG = 64 positions),
// whatever T is -- the first version of this pass (one thread walking row tiles, lanes along g then r) wrote 32-byte runs at T = 32 and ran at 3.4 TB/s.
// Activations per element: (4 JB + 4) / (4 JB) = 1.125 at JB = 8.  ~90 registers: its waves fit beside the two resident GEMM workgroups of another stream.
// =====================================================================================================================================
struct W2dInDev {
    aid_view x; const float* scale; int64_t scale_ld; float mul;      // element = act(x * scale[b,c] * mul)
    float* V;
    // NB mode (the normalisation backward folded into this pass, aid_norm_bwd wform = 3 | 4): the element is gd - coef[b,g] (x - mean[b,g]) + nb_a gy,
    // written to gx for the rows this workgroup owns, and times scale[b,c] into the transform
    aid_view gd, gy, gx; const float* coef; const float* stats; float nb_a; int cg;
    int B, C, F, T, act, dil;
    int R, J, TG, NB;
    int RB, JB, nrb, njb;      // residue classes / row tiles per workgroup, workgroups along each
    int rowf;                  // floats per LDS row group = RB * roww
    int TS, nts, hal, roww;    // T segments (round 6: long rows -- T = 512 / 1024 on the 96-channel levels -- no longer force JB = 1 .. 4): a workgroup takes TS samples
                               // of a row plus `hal` float4 of halo on either side (one sample of each is used); roww = TS + 8 hal floats per class row
    int64_t N;
};

template <int TF, bool NB>
__global__ __launch_bounds__(256) void w2d_input_kernel(const W2dInDev a) {
    constexpr int NTP = TF + 2;                           // planes along T: 6 (F(4,3)) or 10 (F(8,3))
    extern __shared__ __attribute__((aligned(16))) float w2d_slab[];
    int rest = blockIdx.x;
    const int ts = rest % a.nts; rest /= a.nts;
    const int jbk = rest % a.njb; rest /= a.njb;
    const int rbk = rest % a.nrb; rest /= a.nrb;
    const int c = rest % a.C;
    const int b = rest / a.C;
    const int tid = threadIdx.x;
    const int j0 = jbk * a.JB, r0 = rbk * a.RB;
    const int jbe = min(a.JB, a.J - j0);                  // row tiles of this workgroup
    const int nrows = 4 * jbe + 4;                        // sub-lattice indices 4 j0 - 2 .. 4 (j0 + jbe) + 1
    const float sc = (a.scale ? a.scale[(int64_t)b * a.scale_ld + c] : 1.f) * a.mul;
    const float* const xb = a.x.p + (int64_t)b * a.x.sB + (int64_t)c * a.x.sC;
    const int T4 = a.T >> 2, W4 = a.roww >> 2, per_row = a.RB * W4;         // float4s per row / per slab row / per row group
    const int t4base = ts * (a.TS >> 2) - a.hal;          // first float4 of the slab row within the tensor row (the halo float4 first)
    // ---- phase 1: HBM -> activation -> LDS -------------------------------------------------------------------------------------------
    const int n4 = nrows * per_row;
#pragma unroll 4
    for (int e = tid; e < n4; e += 256) {
        const int jl = e / per_row, rem = e - jl * per_row;
        const int rl = rem / W4, q4 = rem - rl * W4;
        const int t4 = t4base + q4;
        const int jj = 4 * j0 - 2 + jl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jj >= 0 && jj < a.R && t4 >= 0 && t4 < T4) {
            const int f = r0 + rl + jj * a.dil;
            v = *reinterpret_cast<const float4*>(xb + (int64_t)f * a.x.sF + 4 * t4);
            if constexpr (NB) {                           // aid_norm_bwd's arithmetic (norm_bwd_kernel), element by element
                const int bg = b * (a.C / a.cg) + c / a.cg;
                const float coef = a.coef[bg], mean = a.stats[2 * bg];
                const float4 g = *reinterpret_cast<const float4*>(a.gd.p + (int64_t)b * a.gd.sB + (int64_t)c * a.gd.sC + (int64_t)f * a.gd.sF + 4 * t4);
                v = make_float4(g.x - coef * (v.x - mean), g.y - coef * (v.y - mean), g.z - coef * (v.z - mean), g.w - coef * (v.w - mean));
                if (a.gy.p) {
                    const float4 y = *reinterpret_cast<const float4*>(a.gy.p + (int64_t)b * a.gy.sB + (int64_t)c * a.gy.sC + (int64_t)f * a.gy.sF + 4 * t4);
                    v.x += a.nb_a * y.x; v.y += a.nb_a * y.y; v.z += a.nb_a * y.z; v.w += a.nb_a * y.w;
                }
                if (jl >= 2 && jl < 2 + 4 * jbe && q4 >= a.hal && q4 < W4 - a.hal)      // the rows / samples of THIS workgroup's tiles (the halo belongs to its neighbours)
                    *reinterpret_cast<float4*>(a.gx.p + (int64_t)b * a.gx.sB + (int64_t)c * a.gx.sC + (int64_t)f * a.gx.sF + 4 * t4) = v;
            }
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            if (a.act == 1) { v.x = aid_gelu(v.x); v.y = aid_gelu(v.y); v.z = aid_gelu(v.z); v.w = aid_gelu(v.w); }      // (gelu(0) = 0: the padding stays zero)
        }
        *reinterpret_cast<float4*>(w2d_slab + 4 * e) = v;
    }
    __syncthreads();
    // ---- phase 2: tiles ----------------------------------------------------------------------------------------------------------------
    const int TGs = a.TS / TF;                            // groups of this segment
    const int per_j = a.RB * TGs;
    const int ntile = jbe * per_j;
    const int64_t pstride = (int64_t)a.C * a.N;
    for (int tl = tid; tl < ntile; tl += 256) {
        const int jl = tl / per_j, rem = tl - jl * per_j;
        const int rl = rem / TGs, gl = rem - rl * TGs;
        const int g = ts * TGs + gl;
        const float* sp = w2d_slab + (4 * jl) * a.rowf + rl * a.roww + 4 * a.hal + TF * gl;
        float W[8][NTP];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* row = sp + i * a.rowf;
            float d[NTP];
            d[0] = (gl > 0 || a.hal) ? row[-1] : 0.f;          // (a halo float4 outside the tensor row was stored as zeros)
#pragma unroll
            for (int q = 0; q < TF / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
                d[1 + 4 * q] = v.x; d[2 + 4 * q] = v.y; d[3 + 4 * q] = v.z; d[4 + 4 * q] = v.w;
            }
            d[TF + 1] = (gl < TGs - 1 || a.hal) ? row[TF] : 0.f;
            if constexpr (TF == 4) aid_w45_input_t(d, W[i]); else aid_wino8_input(d, W[i]);
        }
        float* vp = a.V + (int64_t)c * a.N + (int64_t)b * a.NB + (int64_t)((j0 + jl) * a.dil + r0 + rl) * a.TG + g;
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            float d[8], Vo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = W[i][t];
            aid_w45_input_f(d, Vo);
#pragma unroll
            for (int f = 0; f < 8; ++f) vp[(int64_t)(f * NTP + t) * pstride] = Vo[f];
        }
    }
}

// geometry / blocking of the input pass for a prefilled W2dInDev (x, scale, V, B, C, F, T, act, dil [, the NB fields]) and its launch
static int w2d_input_launch(W2dInDev& a, int TF, bool nb, hipStream_t st) {
    AID_REQUIRE(a.dil >= 1 && (a.F % a.dil) == 0 && (a.T % 16) == 0 && a.T <= 2048, "2-D Winograd input pass: F % dilF == 0, T % 16 == 0, T <= 2048");
    AID_REQUIRE(TF == 4 || (a.T % 32) == 0, "2-D Winograd input pass with F(8,3) along T: T % 32 == 0");
    AID_REQUIRE((a.x.sB % 4) == 0 && (a.x.sC % 4) == 0 && (a.x.sF % 4) == 0 && (((uintptr_t)a.x.p) & 15) == 0, "2-D Winograd input pass: 16-byte aligned input rows");
    const W2dGeo ge = w2d_geo(a.B, a.F, a.T, a.dil, TF);
    AID_REQUIRE(ge.N < (1LL << 31), "2-D Winograd input pass: too many positions");
    a.R = ge.R; a.J = ge.J; a.TG = ge.TG; a.NB = ge.NB; a.N = ge.N;
    // T segments of 512 samples for rows longer than that (one float4 of halo on either side): T = 1024 on level 1 left one row tile per workgroup (eight slab
    // rows for four of output: 3.2 TB/s); segments of 256 measured 4.0 TB/s there but cost T = 512 its 256-byte store runs (4.4 -> 4.0), so 512 it is
    a.TS = (a.T > 512 && (a.T % 512) == 0) ? 512 : a.T;
    a.nts = a.T / a.TS; a.hal = a.nts > 1 ? 1 : 0; a.roww = a.TS + 8 * a.hal;
    const int TGs = a.TS / TF;
    // RB classes per workgroup: 64 positions (256 bytes) per row tile where the dilation has that many classes; the largest divisor of dil below that otherwise
    // (and fewer while a row group would pass 384 floats: the slab must hold at least six row tiles)
    int RB = TGs >= 64 ? 1 : 64 / TGs;
    if (RB > a.dil) RB = a.dil;
    while (a.dil % RB) --RB;
    while (RB > 1 && RB * a.roww > 384) { --RB; while (a.dil % RB) --RB; }
    a.RB = RB; a.nrb = a.dil / RB; a.rowf = RB * a.roww;
    // JB row tiles per workgroup: 8 (activations 1.125 x), fewer when the slab would pass 40 KB (four workgroups per CU; round 6: the F(8,3) form doubles RB at
    // the same T, and a 64-KB slab left two workgroups per CU -- 248 instead of 138 us on [128, 256, 256] d8) or the launch would have too few workgroups;
    // then BALANCED over the row tiles (J = 8 as 7 + 1 loaded 32 + 8 rows for 8 tiles: now 4 + 4)
    // and MORE than 8 where a row tile has few positions (RB TG < 32: T = 32 .. 64 at small dilations), so that phase 2 has a tile for every thread
    // (F(8,3) on [256, 448, 32] d1: 4 positions per row tile left 32 of 256 threads busy -- 63 us against 44 for the F(4,3) form's 64)
    int JBmax = 10240 / a.rowf / 4 - 1;
    if (JBmax < 1) JBmax = 16384 / a.rowf / 4 - 1;         // (very long rows: up to 64 KB)
    int JB = aid_cdiv(256, RB * TGs);
    if (JB < 8) JB = 8;
    if (JB > JBmax) JB = JBmax;
    if (JB > ge.J) JB = ge.J;
    while (JB > 2 && (int64_t)a.B * a.C * a.nrb * a.nts * aid_cdiv(ge.J, JB) < 2048) JB >>= 1;
    AID_REQUIRE(JB >= 1, "2-D Winograd input pass: T too long for the LDS slab");
    JB = aid_cdiv(ge.J, aid_cdiv(ge.J, JB));
    a.JB = JB; a.njb = aid_cdiv(ge.J, JB);
    const int64_t nblk = (int64_t)a.B * a.C * a.nrb * a.njb * a.nts;
    AID_REQUIRE(nblk < (1LL << 31), "2-D Winograd input pass: grid too large");
    const size_t lds = (size_t)(4 * JB + 4) * a.rowf * 4;
    const dim3 grid((unsigned)nblk);
    if (TF == 8) { if (nb) hipLaunchKernelGGL((w2d_input_kernel<8, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((w2d_input_kernel<8, false>), grid, dim3(256), lds, st, a); }
    else { if (nb) hipLaunchKernelGGL((w2d_input_kernel<4, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((w2d_input_kernel<4, false>), grid, dim3(256), lds, st, a); }
    AID_CHECK_LAUNCH();
    return AID_OK;
}

// called by aid_scale_act (aid_norm.hip) for wino == 3 (F(4,3) along T) and wino == 4 (F(8,3) along T)
int aid_w2d_input(const aid_scale_act_params* p, hipStream_t st) {
    W2dInDev a = {};
    a.x = p->x; a.scale = p->scale; a.scale_ld = p->scale_ld; a.V = p->y.p; a.mul = p->mul != 0.f ? p->mul : 1.f;
    a.B = p->B; a.C = p->C; a.F = p->F; a.T = p->T; a.act = p->act; a.dil = p->dilF;
    return w2d_input_launch(a, w2d_tf_of(p->wino), false, st);
}

// called by aid_norm_bwd (aid_norm.hip) for wform == 3 | 4: the normalisation backward AND the 2-D input transform of (its result times wscale[b,c]) for the
// dgrad conv of the layer below in one pass -- reads gd, x, gy once (1.125 x with the halo rows), writes out (= dL/dx) and V
int aid_w2d_input_nb(const aid_norm_bwd_params* p, const float* coef, hipStream_t st) {
    AID_REQUIRE(!p->accumulate && p->wdil >= 1, "aid_norm_bwd(wform=3|4): accumulate = 0 and wdil = the dilation of the layer that reads V");
    AID_REQUIRE((p->gd.sB % 4) == 0 && (p->gd.sC % 4) == 0 && (p->gd.sF % 4) == 0 && (((uintptr_t)p->gd.p) & 15) == 0 &&
                (p->out.sB % 4) == 0 && (p->out.sC % 4) == 0 && (p->out.sF % 4) == 0 && (((uintptr_t)p->out.p) & 15) == 0 &&
                (!p->gy.p || ((p->gy.sB % 4) == 0 && (p->gy.sC % 4) == 0 && (p->gy.sF % 4) == 0 && (((uintptr_t)p->gy.p) & 15) == 0)),
                "aid_norm_bwd(wform=3|4): 16-byte aligned rows");
    W2dInDev a = {};
    a.x = p->x; a.scale = p->wscale; a.scale_ld = p->wscale_ld; a.V = p->wout.p; a.mul = 1.f;
    a.B = p->B; a.C = p->C; a.F = p->F; a.T = p->T; a.act = 0; a.dil = p->wdil;
    a.gd = p->gd; a.gy = p->gy; a.gx = p->out; a.coef = coef; a.stats = p->stats; a.nb_a = p->a; a.cg = p->C / p->groups;
    return w2d_input_launch(a, p->wform == 4 ? 8 : 4, true, st);
}

// =====================================================================================================================================
// 3. output pass: y = AF^T M AT, then the epilogue of the fused kernels.  One thread owns four consecutive sample groups of one row tile and
// output channel (one float4 of every plane): it streams the 8 row planes of the 6 T planes (48 float4 loads), accumulating the four output rows
// in the T-plane domain (24 float4), applies the T transform per row and writes 16 consecutive samples of each of the 4 rows.
//     y = alpha * (res_scale * res + acc * out_scale[b,co])          (epi = 0)
//     y = alpha * (acc * out_scale * gelu'(aux * aux_scale) + ..)    (epi = 1)
// Optional per-block partials: (sum y, sum y^2) -> stat_ws, <y, aux> -> dot_ws, one per (sample, channel, block of 512 positions), in the layout
// aid_group_stats(ws_n) / aid_norm_bwd(ws_n) fold.
// =====================================================================================================================================
struct W2dOutDev {
    aid_conv2d_params p;
    const float* M;
    int R, J, TG, NB, nblk;
    int64_t N;
    int fin_total;             // fin_mode: blocks per sample = Cout * nblk -- the block that finds fin_count[b] == fin_total - 1 folds the sample's partials
};

// One thread owns TWO consecutive sample groups of one row tile and output channel (one float2 of every plane): per row-axis index xf it loads the six
// T planes and applies the T transform first (6 -> 4 samples per group), then accumulates the four output rows with the row-axis coefficients
// (the +- pairs of AF^T: s = a + b, t = a - b) -- 32 accumulators instead of the 96 of a row-transform-first order, ~90 registers, so that eight
// waves per SIMD keep loads in flight and the pass's waves fit beside the two resident GEMM workgroups of another stream.
// Groups per thread GP: TF = 4 two (one float2 of every plane, 135 registers); TF = 8 (F(8,3) along T) ONE group of eight samples (4-byte loads of the 80
// planes; the two-group form needs 240 registers) -- W2D_GP8_TWO (experiment builds, tools/w2d_tf_probe.py) gives rows of at least that many groups two.
#ifndef W2D_GP8_TWO
#define W2D_GP8_TWO 0
#endif
static inline int w2d_gp(int TF, int TG) { return TF == 4 ? 2 : ((W2D_GP8_TWO > 0 && TG >= W2D_GP8_TWO && (TG % 2) == 0) ? 2 : 1); }
// FOLD: M holds the 4 NTP planes j * NTP + xt the folded GEMM wrote (w2d_gemm_fold_kernel): only the T transform is left.
template <int TF, int GP, bool FOLD>
__global__ __launch_bounds__(256) void w2d_output_kernel(const W2dOutDev a) {
    constexpr int NTP = TF + 2;                           // planes along T
    constexpr int NS = GP * TF;                           // output samples per row of this thread
    const aid_conv2d_params& p = a.p;
    int rest = blockIdx.x;
    const int blk = rest % a.nblk; rest /= a.nblk;
    const int co = rest % p.Cout;
    const int b = rest / p.Cout;
    const int tid = threadIdx.x;
    const int np = (blk * 256 + tid) * GP;                // first of this thread's GP positions within the sample
    const bool live = np < a.NB;
    float s1 = 0.f, s2 = 0.f, sd = 0.f;
    if (live) {
        const int q = np / a.TG, g0 = np - q * a.TG;       // row tile index q = j * dil + r; groups g0, g0 + 1 (TG is even)
        const int j = q / p.dilF, r = q - j * p.dilF;
        const float* mp = a.M + ((int64_t)co * a.N + (int64_t)b * a.NB + np);
        const int64_t pstride = (int64_t)p.Cout * a.N;
        float Y[4][NS];
        auto plane = [&](int f, float* o) {                // T transform of row-axis plane f: GP groups x TF samples
            if constexpr (GP == 2) {
                float2 m[NTP];
#pragma unroll
                for (int t = 0; t < NTP; ++t) m[t] = *reinterpret_cast<const float2*>(mp + (int64_t)(f * NTP + t) * pstride);
                float Mt[NTP];
#pragma unroll
                for (int t = 0; t < NTP; ++t) Mt[t] = m[t].x;
                if constexpr (TF == 4) aid_w45_output_t(Mt, o); else aid_wino8_output(Mt, o);
#pragma unroll
                for (int t = 0; t < NTP; ++t) Mt[t] = m[t].y;
                if constexpr (TF == 4) aid_w45_output_t(Mt, o + TF); else aid_wino8_output(Mt, o + TF);
            } else {
                float Mt[NTP];
#pragma unroll
                for (int t = 0; t < NTP; ++t) Mt[t] = mp[(int64_t)(f * NTP + t) * pstride];
                if constexpr (TF == 4) aid_w45_output_t(Mt, o); else aid_wino8_output(Mt, o);
            }
        };
        if constexpr (FOLD) {
            plane(0, Y[0]); plane(1, Y[1]); plane(2, Y[2]); plane(3, Y[3]);
        } else {
        {
            float a0[NS], a1[NS], a2[NS];
            plane(0, a0); plane(1, a1); plane(2, a2);
#pragma unroll
            for (int e = 0; e < NS; ++e) { const float s = a1[e] + a2[e], t = a1[e] - a2[e]; Y[0][e] = a0[e] + s; Y[1][e] = t; Y[2][e] = s; Y[3][e] = t; }
        }
        {
            float a1[NS], a2[NS];
            plane(3, a1); plane(4, a2);
#pragma unroll
            for (int e = 0; e < NS; ++e) { const float s = a1[e] + a2[e], t = a1[e] - a2[e]; Y[0][e] += s; Y[1][e] += 2.0f * t; Y[2][e] += 4.0f * s; Y[3][e] += 8.0f * t; }
        }
        {
            float a1[NS], a2[NS], a7[NS];
            plane(5, a1); plane(6, a2); plane(7, a7);
#pragma unroll
            for (int e = 0; e < NS; ++e) { const float s = a1[e] + a2[e], t = a1[e] - a2[e]; Y[0][e] += s; Y[1][e] += 0.5f * t; Y[2][e] += 0.25f * s; Y[3][e] += 0.125f * t + a7[e]; }
        }
        }
        const float sv = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + co] : 1.f;
        const float as = p.epi == 1 ? p.aux_scale[(int64_t)b * p.aux_scale_ld + co] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jj = 4 * j + i;
            if (jj >= a.R) continue;
            const int f = r + jj * p.dilF;
            const int t0 = TF * g0;
            float* yp = p.y.p + (int64_t)b * p.y.sB + (int64_t)co * p.y.sC + (int64_t)f * p.y.sF + t0;
            float rr[NS], uu[NS];
            if (p.res.p) {
                const float* rp = p.res.p + (int64_t)b * p.res.sB + (int64_t)co * p.res.sC + (int64_t)f * p.res.sF + t0;
#pragma unroll
                for (int v4 = 0; v4 < NS / 4; ++v4) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + 4 * v4);
                    rr[4 * v4] = v.x; rr[4 * v4 + 1] = v.y; rr[4 * v4 + 2] = v.z; rr[4 * v4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e) rr[e] = 0.f;
            }
            if (p.aux.p) {
                const float* up = p.aux.p + (int64_t)b * p.aux.sB + (int64_t)co * p.aux.sC + (int64_t)f * p.aux.sF + t0;
#pragma unroll
                for (int v4 = 0; v4 < NS / 4; ++v4) {
                    const float4 v = *reinterpret_cast<const float4*>(up + 4 * v4);
                    uu[4 * v4] = v.x; uu[4 * v4 + 1] = v.y; uu[4 * v4 + 2] = v.z; uu[4 * v4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e) uu[e] = 0.f;
            }
            float y[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                float v = Y[i][e] * sv;
                if (p.epi == 1) v *= aid_dgelu(uu[e] * as);
                v += p.res_scale * rr[e];
                v *= p.alpha;
                y[e] = v;
                s1 += v; s2 += v * v; sd += v * uu[e];
            }
#pragma unroll
            for (int v4 = 0; v4 < NS / 4; ++v4) *reinterpret_cast<float4*>(yp + 4 * v4) = make_float4(y[4 * v4], y[4 * v4 + 1], y[4 * v4 + 2], y[4 * v4 + 3]);
        }
    }
    if (!p.stat_ws && !p.dot_ws) return;
    // block partials (fixed order: lanes by xor tree, then the four waves)
    double d1 = (double)s1, d2 = (double)s2, d3 = (double)sd;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); d3 += __shfl_xor(d3, off, 64); }
    __shared__ double red[4][3];
    if ((tid & 63) == 0) { red[tid >> 6][0] = d1; red[tid >> 6][1] = d2; red[tid >> 6][2] = d3; }
    __syncthreads();
    if (tid == 0) {
        const int cpg = p.Cout >> 3;
        const int grp = co / cpg;
        const int slot = (co - grp * cpg) * a.nblk + blk;
        const int nslot = cpg * a.nblk;
        if (p.stat_ws) {
            double* w = p.stat_ws + (((int64_t)b * 8 + grp) * nslot + slot) * 2;
            const double v0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), v1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
            if (p.fin_mode) { aid_st_agent(w, v0); aid_st_agent(w + 1, v1); }
            else { w[0] = v0; w[1] = v1; }
        }
        if (p.dot_ws) {
            double* w = p.dot_ws + ((int64_t)b * 8 + grp) * nslot + slot;
            const double v = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
            if (p.fin_mode) aid_st_agent(w, v);
            else *w = v;
        }
    }
    if (p.fin_mode) {                                        // (block-uniform) the last block of sample b folds the sample's partials (aid_fin.h), as the
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // row-shared kernels do: this block's partials have reached the coherent level,
        __syncthreads();                                      // then ONE arrival per block
        __shared__ int last;
        __shared__ double fsh[16];
        if (tid == 0) last = (__hip_atomic_fetch_add(p.fin_count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.fin_total - 1)) ? 1 : 0;
        __syncthreads();
        if (last) {
            const double n = (double)(p.Cout >> 3) * (double)p.F * (double)p.T;
            aid_wino_fin(p.fin_mode, p.stat_ws ? p.stat_ws : p.dot_ws, (p.Cout >> 3) * a.nblk, b, p.Cout, n, p.fin_gamma, p.fin_mod, p.fin_mod_ld, p.fin_eps,
                         p.fin_scale, p.fin_stats, fsh, tid, 256);
            if (tid == 0) __hip_atomic_store(p.fin_count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // as found, for the next launch
        }
    }
}

static inline int w2d_nblk(int NB, int TF, int TG) { return (NB + 256 * w2d_gp(TF, TG) - 1) / (256 * w2d_gp(TF, TG)); }      // blocks of 256 threads x GP positions per sample and channel

extern "C" int aid_conv2d_wino2d_supported(int Cin, int Cout, int F, int T, int dilF) { return w2d_shape_ok(Cin, Cout, F, T, dilF) ? 1 : 0; }
// Which launches take the 2-D form instead of the fused 1-D kernels.  Measured per layer (tools/w2d_probe.py layer, profiles/r05_w2d_passes_ab.txt,
// the three passes together against the fused F(8,3) / F(4,3) kernel with its pre-pass):
//   K = Cout = 256 (levels 5, 6): 0.61-0.74 at batch 4 (the GEMM runs at 0.77-0.85 of the fp32 MFMA peak on 2.08x fewer products; the two transform
//     passes move 9 x the activation at 4.4-7 TB/s), 0.55-0.58 at batch 1 (48 x more tiles than the fused kernel has to fill the chip with);
//   K = 128: the GEMM sits at the HBM ridge (32 FLOP/B).  T <= 128 (level 4, the 128-channel step of level 5): 0.82-0.94 at batch 4, 0.58-0.63 at
//     batch 1; T = 256 (level 3): 1.02 at batch 4 per layer ALONE, 0.87 at batch 1 -- and +1 % END TO END at batch 8 (three alternating runs: 55.49 ->
//     55.9 evaluations/s, profiles/r05_w2d_force_ab.txt; with the K = 128 levels altogether 53.6 -> 55.3, r05_w2d_k128_ab.txt): beside the other
//     sub-batch stream the form with fewer MFMAs wins where the isolated A/B ties.  Taken up to T = 256.
// Ragged row tiles (F / dil not a multiple of 4) pad the GEMM by up to 4/3: K = 256 still wins at 1.33 (0.84), K = 128 only up to 1.2.
// A function of the launch shape, B included: equal launches (the same sub-batch size) take equal kernels.
extern "C" int aid_conv2d_wino2d_wanted(int B, int Cin, int Cout, int F, int T, int dilF) {
    if (!w2d_shape_ok(Cin, Cout, F, T, dilF)) return 0;
    const W2dGeo ge = w2d_geo(B, F, T, dilF);
    const double pad = (double)ge.J * 4.0 / (double)ge.R;
    // round 6: the 96-channel levels (1, 2 and the decoder of 3) on the 80-plane form with 96 x 128 GEMM tiles -- with 48 planes this lost 0.6-1.2 % end to end (round 5),
    // with 80 it is 0.87-0.94 of the fused F(8,3) path per layer at T <= 512 and ties at T = 1024 (profiles/r06_w2d_c96_tf8_probe.txt: batch 1 0.77-0.98), and end to end
    // 60.5 / 60.7 -> 62.1 / 61.6 evaluations/s at batch 8, 38.4 -> 39.8 at batch 1, 48.2 -> 49.9 at batch 2 (profiles/r06_w2d_c96_bench_ab.txt)
    if ((Cout % 128) != 0) return (Cin >= 96 && (T % 32) == 0 && T <= 1024 && pad <= 1.21) ? 1 : 0;
    if (Cin >= 256 && Cout >= 256) return pad <= 1.34 ? 1 : 0;
    if (Cin < 128 || Cout < 128 || pad > 1.21) return 0;
    if (T <= 256) return 1;
    return (int64_t)B * F * T <= 2 * 65536 ? 1 : 0;
}
extern "C" int aid_conv2d_wino2d_wanted(int B, int Cin, int Cout, int F, int T, int dilF);
extern "C" int64_t aid_conv2d_wino2d_positions(int B, int F, int T, int dilF) { return (dilF >= 1 && F % dilF == 0 && T % 4 == 0) ? w2d_geo(B, F, T, dilF).N : 0; }
// per-(sample, group) partial count of stat_ws / dot_ws for x_wino = 3
int aid_w2d_partials(int Cout, int F, int T, int dilF, int TF) { const W2dGeo g = w2d_geo(1, F, T, dilF, TF); return (Cout >> 3) * w2d_nblk(g.NB, TF, g.TG); }

// Which T form the 2-D form of a launch should take: 0 = the launch is not for the 2-D form at all (aid_conv2d_wino2d_wanted == 0), 4 = F(4,5) x F(4,3)
// (x_wino = 3, 48 planes), 8 = F(4,5) x F(8,3) (x_wino = 4, 80 planes: 2.5 instead of 3.0 products per output and 2.5 x instead of 3 x the activation in V / M).
// A function of the launch shape, B included.
extern "C" int aid_conv2d_wino2d_tform(int B, int Cin, int Cout, int F, int T, int dilF) {
    if (!aid_conv2d_wino2d_wanted(B, Cin, Cout, F, T, dilF)) return 0;
    if (T % 32) return 4;                                    // (T / 8 groups per row, a multiple of 4: the GEMM's N % 4 == 0)
    // per layer (tools/w2d_tf_probe.py, profiles/r06_w2d_tf8_layer_ab.txt): the three passes of the F(8,3) form take 0.81-0.92 of the F(4,3) form's time at
    // batch 4 and 8 on every level, 0.87-0.91 on level 5 at batch 1; a launch with fewer than 512 positions per plane (level 6 at batch 1: 448, the last GEMM
    // column tile 7/8 empty) ties (0.98-1.03) and keeps the more accurate form
    return w2d_geo(B, F, T, dilF, 8).N >= 512 ? 8 : 4;
}

// aid_conv2d with x_wino = 3: x.p = V [48][Cin][N], wp_wino = U [48][Cin_pad][Cout_pad], ws = scratch for M [48][Cout][N]
static int w2d_check(const aid_conv2d_params* p) {
    const int TF = w2d_tf_of(p->x_wino), NXI = 8 * (TF + 2);
    AID_REQUIRE(p->KH == 5 && p->KW == 3 && p->wp_wino && p->wino_taps == NXI, "aid_conv2d(x_wino=3|4): needs the 48- / 80-plane pack (aid_pack_conv_weight wpw2 / wpw3)");
    AID_REQUIRE(w2d_shape_ok(p->Cin, p->Cout, p->F, p->T, p->dilF), "aid_conv2d(x_wino=3): shape not supported (aid_conv2d_wino2d_supported)");
    AID_REQUIRE(!p->in_scale && p->act == 0 && !p->x2.p, "aid_conv2d(x_wino=3): no in-kernel prologue, no x2");
    if (p->fin_mode) {                                       // the last block of a sample folds the partials (aid_conv2d_fin_supported(.., 3) == 1)
        AID_REQUIRE(p->fin_mode == 1 || p->fin_mode == 2, "aid_conv2d: fin_mode is 0, 1 or 2");
        AID_REQUIRE(p->fin_count && p->fin_scale && (p->Cout % 8) == 0, "aid_conv2d: fin_mode needs fin_count, fin_scale and Cout % 8 == 0");
        if (p->fin_mode == 1) AID_REQUIRE(p->stat_ws && p->fin_gamma, "aid_conv2d: fin_mode = 1 folds the stat_ws partials and needs fin_gamma");
        else AID_REQUIRE(p->dot_ws && p->fin_stats, "aid_conv2d: fin_mode = 2 folds the dot_ws partials and needs the forward statistics in fin_stats");
    }
    AID_REQUIRE(!(p->stat_ws && p->dot_ws), "aid_conv2d(x_wino=3): stat_ws and dot_ws are exclusive");
    AID_REQUIRE(!p->dot_ws || p->epi == 1, "aid_conv2d(x_wino=3): dot_ws goes with the dGELU epilogue");
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF, TF);
    AID_REQUIRE(p->ws && p->ws_bytes >= (int64_t)NXI * p->Cout * ge.N * 4 && (((uintptr_t)p->ws) & 15) == 0, "aid_conv2d(x_wino=3|4): ws must hold 48 (80) * Cout * N floats");
    AID_REQUIRE((p->y.sB % 4) == 0 && (p->y.sC % 4) == 0 && (p->y.sF % 4) == 0 && (((uintptr_t)p->y.p) & 15) == 0, "aid_conv2d(x_wino=3): 16-byte aligned output rows");
    AID_REQUIRE(!p->res.p || ((p->res.sB % 4) == 0 && (p->res.sC % 4) == 0 && (p->res.sF % 4) == 0 && (((uintptr_t)p->res.p) & 15) == 0), "aid_conv2d(x_wino=3): 16-byte aligned residual rows");
    AID_REQUIRE(!p->aux.p || ((p->aux.sB % 4) == 0 && (p->aux.sC % 4) == 0 && (p->aux.sF % 4) == 0 && (((uintptr_t)p->aux.p) & 15) == 0), "aid_conv2d(x_wino=3): 16-byte aligned aux rows");
    const int npart = aid_w2d_partials(p->Cout, p->F, p->T, p->dilF, TF);
    AID_REQUIRE(!p->stat_ws || p->stat_n == npart, "aid_conv2d(x_wino=3): stat_n != aid_conv2d_stat_partials()");
    AID_REQUIRE(!p->dot_ws || p->dot_n == npart, "aid_conv2d(x_wino=3): dot_n != aid_conv2d_dot_partials()");
    return AID_OK;
}
// Which launches fold the row-axis output transform into the GEMM (M' with 4 NTP planes): the HBM-bound ones -- Cin <= 128 on the 80-plane form.
// A function of the layer shape; the GEMM and the output pass of a launch both ask it.  (W2D_FOLD_M = 0: never -- experiment builds for the A/B.)
#ifndef W2D_FOLD_M
#define W2D_FOLD_M 1
#endif
#ifndef W2D_FOLD_MAX_CIN
#define W2D_FOLD_MAX_CIN 128
#endif
#ifndef W2D_FOLD_WPC
#define W2D_FOLD_WPC 2
#endif
#ifndef W2D_FOLD_MIN_WGS
#define W2D_FOLD_MIN_WGS 768
#endif
#ifndef W2D_FOLD96_KC          // (experiment builds: K chunk and buffer count of the three-wave 96 x 64 instance)
#define W2D_FOLD96_KC 24
#endif
#ifndef W2D_FOLD96_NBUF
#define W2D_FOLD96_NBUF 3
#endif
#define W2D_STR2(x) #x
#define W2D_STR(x) W2D_STR2(x)
static inline bool w2d_fold_m(const aid_conv2d_params* p) {
    if (!(W2D_FOLD_M && p->x_wino == 4 && p->Cin <= W2D_FOLD_MAX_CIN && (p->Cin % 16) == 0 && ((p->Cout_pad % 128) == 0 || (p->Cout_pad % 96) == 0))) return false;
    // the folded GEMM has an eighth of the workgroups (each walks eight planes) on 64-column tiles: it needs a launch that still fills the chip a few times --
    // per layer it wins 4-9 % at >= 1280 workgroups (level 3 at the product's sub-batch of four), ties at 800 and loses 10-50 % below 650 (batch 1;
    // profiles/r06_w2d_foldm_layer_ab.txt); end to end (profiles/r06_w2d_foldm_threshold_ab.txt, configs[1], alternating runs): no fold 61.48 / 61.52,
    // threshold 1900 61.63 / 61.41, 1024 62.38 / 62.04, 768 62.89 / 62.55 evaluations/s; batch 1 / 2 and configs[4] unchanged under 768
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF, 8);
    const int64_t wgs = (int64_t)10 * aid_cdiv(ge.N, 64) * aid_cdiv(p->Cout, 128);
    return wgs >= W2D_FOLD_MIN_WGS;
}
static int w2d_gemm_of(const aid_conv2d_params* p, hipStream_t st) {
    const int TF = w2d_tf_of(p->x_wino);
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF, TF);
    aid_wino2d_gemm_params gp;
    gp.U = p->wp_wino; gp.V = p->x.p; gp.M = p->ws;
    gp.nxi = 8 * (TF + 2); gp.Cin = p->Cin; gp.Cout = p->Cout; gp.Cin_pad = p->Cin_pad; gp.Cout_pad = p->Cout_pad; gp.N = ge.N; gp.variant = 0;
    if (w2d_fold_m(p)) {
        AID_REQUIRE((gp.N % 4) == 0, "aid_conv2d(x_wino=4): N % 4 == 0");
        // The instances (profiles/r06_w2d_fold_tiles_ab.txt, per launch in the evaluation): 128-channel panels on four waves x (32 x 64), K chunks of 32 where Cin allows
        // (276 -> 264 us against chunks of 16: half the barriers; more buffers of 16 measured nothing, the loads are not what it waits for); 96-channel panels on THREE
        // waves with K chunks of 24 (exact 96 x 64 tiles, 15 one-KiB pieces per chunk = 5 per wave): 409 us against 437 on the four-wave instance with a quarter
        // of its columns clamped and 500 on two waves x (96 x 32).  Other Cin (multiples of 16) keep the four-wave instance with chunks of 16.
        const bool p96 = (p->Cout_pad % 128) != 0;
        if (p96 && (p->Cin % W2D_FOLD96_KC) == 0)
            return w2d_launch_gemm_fold<1, 2, 3, 1, W2D_FOLD96_KC, W2D_FOLD96_NBUF, W2D_FOLD_WPC>(&gp, TF + 2, st, "w2d_gemm_kernel<96x64,foldM,kc" W2D_STR(W2D_FOLD96_KC) ",nb" W2D_STR(W2D_FOLD96_NBUF) ">+t8");
        if ((p->Cin % 32) == 0)
            return w2d_launch_gemm_fold<1, 2, 4, 1, 32, 3, W2D_FOLD_WPC>(&gp, TF + 2, st, p96 ? "w2d_gemm_kernel<128x64(96),foldM,kc32,nb3>+t8" : "w2d_gemm_kernel<128x64,foldM,kc32,nb3>+t8");
        return w2d_launch_gemm_fold<1, 2, 4, 1, 16, 3, W2D_FOLD_WPC>(&gp, TF + 2, st, p96 ? "w2d_gemm_kernel<128x64(96),foldM,kc16,nb3>+t8" : "w2d_gemm_kernel<128x64,foldM,kc16,nb3>+t8");
    }
    return aid_wino2d_gemm(&gp, st);
}
static int w2d_output_of(const aid_conv2d_params* p, hipStream_t st) {
    const int TF = w2d_tf_of(p->x_wino);
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF, TF);
    W2dOutDev a;
    a.p = *p; a.M = p->ws;
    a.R = ge.R; a.J = ge.J; a.TG = ge.TG; a.NB = ge.NB; a.N = ge.N; a.nblk = w2d_nblk(ge.NB, TF, ge.TG);
    a.fin_total = p->Cout * a.nblk;
    const dim3 grid((unsigned)((int64_t)p->B * p->Cout * a.nblk));
    if (TF == 8 && w2d_gp(8, ge.TG) == 2) hipLaunchKernelGGL((w2d_output_kernel<8, 2, false>), grid, dim3(256), 0, st, a);
    else if (TF == 8 && w2d_fold_m(p)) hipLaunchKernelGGL((w2d_output_kernel<8, 1, true>), grid, dim3(256), 0, st, a);
    else if (TF == 8) hipLaunchKernelGGL((w2d_output_kernel<8, 1, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((w2d_output_kernel<4, 2, false>), grid, dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
int aid_conv53_wino2d(const aid_conv2d_params* p, hipStream_t st) {
    int rc = w2d_check(p);
    if (rc != AID_OK) return rc;
    rc = w2d_gemm_of(p, st);
    if (rc != AID_OK) return rc;
    return w2d_output_of(p, st);
}
// The two launches of aid_conv2d(x_wino = 3) as separate calls on the same parameter block (a launch plan that wants the MFMA-bound GEMM and the
// HBM-bound output pass as separate nodes: per-kernel timing, different streams).  _gemm then _output on one stream == aid_conv2d(p).
extern "C" int aid_conv2d_wino2d_gemm(const aid_conv2d_params* p, void* stream) {
    AID_REQUIRE(p && p->x.p && p->y.p && (p->x_wino == 3 || p->x_wino == 4), "aid_conv2d_wino2d_gemm: x_wino = 3 | 4 parameters");
    const int rc = w2d_check(p);
    return rc != AID_OK ? rc : w2d_gemm_of(p, (hipStream_t)stream);
}
extern "C" int aid_conv2d_wino2d_output(const aid_conv2d_params* p, void* stream) {
    AID_REQUIRE(p && p->x.p && p->y.p && (p->x_wino == 3 || p->x_wino == 4), "aid_conv2d_wino2d_output: x_wino = 3 | 4 parameters");
    const int rc = w2d_check(p);
    if (rc != AID_OK) return rc;
    aid_note_kernel("w2d_output_kernel");
    return w2d_output_of(p, (hipStream_t)stream);
}
