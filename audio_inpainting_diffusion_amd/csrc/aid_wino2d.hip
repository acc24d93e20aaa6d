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
#include "aid_common.h"
#include "aid_wino45.h"
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

template <int MT, int NT, int WGM, int WGN, int KC, int NBUF, int WPC>
static int w2d_launch_gemm(const aid_wino2d_gemm_params* p, hipStream_t st, const char* name) {
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
    aid_note_kernel(name);
    return AID_OK;
}

extern "C" int aid_wino2d_gemm(const aid_wino2d_gemm_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->U && p->V && p->M, "aid_wino2d_gemm: null pointer");
    AID_REQUIRE(p->nxi > 0 && p->Cin > 0 && p->Cout > 0 && p->N > 0 && (p->N % 4) == 0 && p->N < (1LL << 31), "aid_wino2d_gemm: bad shape (N % 4 == 0)");
    AID_REQUIRE(((uintptr_t)p->U & 15) == 0 && ((uintptr_t)p->V & 15) == 0 && ((uintptr_t)p->M & 15) == 0 && (p->Cout_pad % 4) == 0, "aid_wino2d_gemm: 16-byte alignment");
    AID_REQUIRE(p->Cin_pad >= p->Cin && p->Cout_pad >= p->Cout, "aid_wino2d_gemm: padded dims");
    int variant = p->variant;
    if (variant == 0) {
        // 128 x 256 tiles (two workgroups per CU) while the launch has a few rounds of them; 128 x 128 tiles at four workgroups per CU for the short ones
        // (batch 1-2, the deepest level): 15-20 % faster there (profiles/r05_w2d_layer_probe2_b1.txt).  Every output element sums its K products in
        // the same order under either tiling: the choice does not change a bit of the result.
        const int64_t tiles = (int64_t)p->nxi * aid_cdiv(p->N, 256) * aid_cdiv(p->Cout, 128);
        variant = tiles < 1600 ? 11 : 12;
    }
    switch (variant) {
        case 12: return w2d_launch_gemm<2, 4, 2, 2, 16, 3, 2>(p, st, "w2d_gemm_kernel<128x256,kc16,nb3>");
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
        case 11: return w2d_launch_gemm<2, 2, 2, 2, 16, 2, 4>(p, st, "w2d_gemm_kernel<128x128,kc16,nb2,wpc4>");
        default: break;
    }
    aid_set_error("aid_wino2d_gemm: unknown variant");
    return AID_E_BADARG;
}

// =====================================================================================================================================
// Geometry shared by the three passes.  R = F / dil rows per residue class, J = ceil(R / 4) row tiles, TG = T / 4 sample groups,
// NB = J * dil * TG positions per sample, N = B * NB.  Position n' = (j * dil + r) * TG + g.
// =====================================================================================================================================
struct W2dGeo { int R, J, TG, NB; int64_t N; };
static inline W2dGeo w2d_geo(int B, int F, int T, int dil) {
    W2dGeo g;
    g.R = F / dil; g.J = (g.R + 3) / 4; g.TG = T / 4; g.NB = g.J * dil * g.TG; g.N = (int64_t)B * g.NB;
    return g;
}
static inline bool w2d_shape_ok(int Cin, int Cout, int F, int T, int dil) {
    return dil >= 1 && (F % dil) == 0 && (T % 16) == 0 && T >= 16 && (Cin % 16) == 0 && (Cout % 128) == 0 && Cin >= 64;
}

// =====================================================================================================================================
// 1. input pass: V[xi][c][n] = BF^T act(x * scale[b,c]) BT.  One thread walks SEG consecutive row tiles of ONE (sample, channel, residue class,
// sample group): every new tile loads its four new sub-lattice rows (float4 + the two neighbour samples from the adjacent lanes), activates and
// T-transforms them once, and keeps the last four of the previous tile in registers (a window of 8 rows x 6 values); then the row transform per
// T plane and 48 stores.  Lanes run along g, then r: a wave reads whole contiguous rows and writes 256-byte runs of every plane.
// =====================================================================================================================================
struct W2dInDev {
    aid_view x; const float* scale; int64_t scale_ld;
    float* V;
    int B, C, F, T, act, dil;
    int R, J, TG, NB, nseg, seg;
    int64_t N, total;
};

__global__ __launch_bounds__(256) void w2d_input_kernel(const W2dInDev a) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < a.total;
    int64_t rest = live ? gid : a.total - 1;
    const int g = (int)(rest % a.TG); rest /= a.TG;
    const int r = (int)(rest % a.dil); rest /= a.dil;
    const int sg = (int)(rest % a.nseg); rest /= a.nseg;
    const int c = (int)(rest % a.C);
    const int b = (int)(rest / a.C);
    const int lane = threadIdx.x & 63;
    const float sc = a.scale ? a.scale[(int64_t)b * a.scale_ld + c] : 1.f;
    const float* const xb = a.x.p + (int64_t)b * a.x.sB + (int64_t)c * a.x.sC + (int64_t)r * a.x.sF + 4 * g;
    const int64_t rstep = (int64_t)a.dil * a.x.sF;
    const bool first_g = g == 0, last_g = g == a.TG - 1;

    float W[8][6];
    float4 raw[4];
    float hl[4], hr[4];                                   // neighbour samples of the wave's first / last lane (only when a row has more than 64 groups)
    auto fetch = [&](int jbase) {                         // raw rows jbase .. jbase + 3 of this thread's residue class (zeros outside [0, R))
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jj = jbase + i;
            const bool in = jj >= 0 && jj < a.R;
            const float* xr = xb + (int64_t)jj * rstep;
            raw[i] = in ? *reinterpret_cast<const float4*>(xr) : make_float4(0.f, 0.f, 0.f, 0.f);
            hl[i] = (in && lane == 0 && !first_g) ? xr[-1] : 0.f;
            hr[i] = (in && lane == 63 && !last_g) ? xr[4] : 0.f;
        }
    };
    auto process = [&](int i, float* w) {                 // activate and T-transform fetched row i
        float d[6];
        float4 v = raw[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        float lf0 = hl[i] * sc, rt0 = hr[i] * sc;
        if (a.act == 1) {                                 // (gelu(0) = 0: the padding stays zero)
            v.x = aid_gelu(v.x); v.y = aid_gelu(v.y); v.z = aid_gelu(v.z); v.w = aid_gelu(v.w);
            if (lane == 0) lf0 = aid_gelu(lf0);
            if (lane == 63) rt0 = aid_gelu(rt0);
        }
        d[1] = v.x; d[2] = v.y; d[3] = v.z; d[4] = v.w;
        float lf = __shfl_up(v.w, 1), rt = __shfl_down(v.x, 1);             // lane - 1 / lane + 1: groups g - 1 / g + 1 of the same row (g runs fastest)
        if (lane == 0) lf = lf0;
        if (lane == 63) rt = rt0;
        d[0] = first_g ? 0.f : lf;
        d[5] = last_g ? 0.f : rt;
        aid_w45_input_t(d, w);
    };
    const int j0 = sg * a.seg;
    const int j1 = min(a.J, j0 + a.seg);
    fetch(4 * j0 - 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) process(i, W[4 + i]);
    fetch(4 * j0 + 2);
    float* const vb = a.V + (int64_t)c * a.N + (int64_t)b * a.NB + (int64_t)r * a.TG + g;
    const int64_t pstride = (int64_t)a.C * a.N;
    for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 6; ++t) W[i][t] = W[4 + i][t];
#pragma unroll
        for (int i = 0; i < 4; ++i) process(i, W[4 + i]);
        if (j + 1 < j1) fetch(4 * j + 6);                 // the next tile's rows are in flight while this one is transformed and stored
        float* vp = vb + (int64_t)j * a.dil * a.TG;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            float d[8], V[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = W[i][t];
            aid_w45_input_f(d, V);
            if (live) {
#pragma unroll
                for (int f = 0; f < 8; ++f) vp[(int64_t)(f * 6 + t) * pstride] = V[f];
            }
        }
    }
}

// called by aid_scale_act (aid_norm.hip) for wino == 3
int aid_w2d_input(const aid_scale_act_params* p, hipStream_t st) {
    AID_REQUIRE(p->dilF >= 1 && (p->F % p->dilF) == 0 && (p->T % 16) == 0, "aid_scale_act(wino=3): F % dilF == 0 and T % 16 == 0");
    AID_REQUIRE((p->x.sB % 4) == 0 && (p->x.sC % 4) == 0 && (p->x.sF % 4) == 0 && (((uintptr_t)p->x.p) & 15) == 0, "aid_scale_act(wino=3): 16-byte aligned input rows");
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF);
    AID_REQUIRE(ge.N < (1LL << 31), "aid_scale_act(wino=3): too many positions");
    W2dInDev a;
    a.x = p->x; a.scale = p->scale; a.scale_ld = p->scale_ld; a.V = p->y.p;
    a.B = p->B; a.C = p->C; a.F = p->F; a.T = p->T; a.act = p->act; a.dil = p->dilF;
    a.R = ge.R; a.J = ge.J; a.TG = ge.TG; a.NB = ge.NB; a.N = ge.N;
    // tiles per thread: long walks amortise the four warm-up rows, short ones give the chip enough threads (>= ~4 waves per SIMD)
    int seg = 8;
    while (seg > 1 && (int64_t)p->B * p->C * p->dilF * ge.TG * aid_cdiv(ge.J, seg) < 256 * 1024) seg >>= 1;
    a.seg = seg; a.nseg = aid_cdiv(ge.J, seg);
    a.total = (int64_t)p->B * p->C * a.nseg * p->dilF * ge.TG;
    hipLaunchKernelGGL(w2d_input_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
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
};

__global__ __launch_bounds__(128) void w2d_output_kernel(const W2dOutDev a) {
    const aid_conv2d_params& p = a.p;
    int rest = blockIdx.x;
    const int blk = rest % a.nblk; rest /= a.nblk;
    const int co = rest % p.Cout;
    const int b = rest / p.Cout;
    const int tid = threadIdx.x;
    const int np = (blk * 128 + tid) * 4;                 // first of this thread's 4 positions within the sample
    const bool live = np < a.NB;
    float s1 = 0.f, s2 = 0.f, sd = 0.f;
    if (live) {
        const int q = np / a.TG, g0 = np - q * a.TG;       // row tile index q = j * dil + r; groups g0 .. g0+3 (TG % 4 == 0)
        const int j = q / p.dilF, r = q - j * p.dilF;
        const float* mp = a.M + ((int64_t)co * a.N + (int64_t)b * a.NB + np);
        const int64_t pstride = (int64_t)p.Cout * a.N;
        constexpr float AF[4][8] = AID_W45_ATF;
        float4 P[4][6];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 6; ++t) P[i][t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            float4 m[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) m[t] = *reinterpret_cast<const float4*>(mp + (int64_t)(f * 6 + t) * pstride);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float cf = AF[i][f];
                if (cf == 0.f) continue;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    P[i][t].x += cf * m[t].x; P[i][t].y += cf * m[t].y; P[i][t].z += cf * m[t].z; P[i][t].w += cf * m[t].w;
                }
            }
        }
        const float sv = p.out_scale ? p.out_scale[(int64_t)b * p.out_scale_ld + co] : 1.f;
        const float as = p.epi == 1 ? p.aux_scale[(int64_t)b * p.aux_scale_ld + co] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jj = 4 * j + i;
            if (jj >= a.R) continue;
            const int f = r + jj * p.dilF;
            const int t0 = 4 * g0;
            float* yp = p.y.p + (int64_t)b * p.y.sB + (int64_t)co * p.y.sC + (int64_t)f * p.y.sF + t0;
            const float* rp = p.res.p ? p.res.p + (int64_t)b * p.res.sB + (int64_t)co * p.res.sC + (int64_t)f * p.res.sF + t0 : nullptr;
            const float* up = p.aux.p ? p.aux.p + (int64_t)b * p.aux.sB + (int64_t)co * p.aux.sC + (int64_t)f * p.aux.sF + t0 : nullptr;
            float4 rv[4], uv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rv[k] = rp ? *reinterpret_cast<const float4*>(rp + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
                uv[k] = up ? *reinterpret_cast<const float4*>(up + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                  // group g0 + k: component k of the float4s
                float Mt[6], y[4];
#pragma unroll
                for (int t = 0; t < 6; ++t) Mt[t] = w2d_get(P[i][t], k);
                aid_w45_output_t(Mt, y);
                const float rr[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
                const float uu[4] = {uv[k].x, uv[k].y, uv[k].z, uv[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = y[e] * sv;
                    if (p.epi == 1) v *= aid_dgelu(uu[e] * as);
                    v += p.res_scale * rr[e];
                    v *= p.alpha;
                    y[e] = v;
                    s1 += v; s2 += v * v; sd += v * uu[e];
                }
                *reinterpret_cast<float4*>(yp + 4 * k) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
    if (!p.stat_ws && !p.dot_ws) return;
    // block partials (fixed order: lanes by xor tree, then the two waves)
    double d1 = (double)s1, d2 = (double)s2, d3 = (double)sd;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); d3 += __shfl_xor(d3, off, 64); }
    __shared__ double red[2][3];
    if ((tid & 63) == 0) { red[tid >> 6][0] = d1; red[tid >> 6][1] = d2; red[tid >> 6][2] = d3; }
    __syncthreads();
    if (tid == 0) {
        const int cpg = p.Cout >> 3;
        const int grp = co / cpg;
        const int slot = (co - grp * cpg) * a.nblk + blk;
        const int nslot = cpg * a.nblk;
        if (p.stat_ws) {
            double* w = p.stat_ws + (((int64_t)b * 8 + grp) * nslot + slot) * 2;
            w[0] = red[0][0] + red[1][0];
            w[1] = red[0][1] + red[1][1];
        }
        if (p.dot_ws) p.dot_ws[((int64_t)b * 8 + grp) * nslot + slot] = red[0][2] + red[1][2];
    }
}

static inline int w2d_nblk(int NB) { return (NB / 4 + 127) / 128; }

extern "C" int aid_conv2d_wino2d_supported(int Cin, int Cout, int F, int T, int dilF) { return w2d_shape_ok(Cin, Cout, F, T, dilF) ? 1 : 0; }
// Which launches take the 2-D form instead of the fused 1-D kernels.  Measured per layer (tools/w2d_probe.py layer, profiles/r05_w2d_layer_probe*.txt):
// at K = Cout = 256 the three passes together take 0.67-0.81 of the fused F(8,3) / F(4,3) kernel with its pre-pass (the GEMM runs at 0.80-0.85 of
// the fp32 MFMA peak on 2.08x fewer products; the two transform passes move 9 x the activation at 3.5-6 TB/s); at K = 128 the GEMM sits at the
// HBM ridge (32 FLOP/B) and the sum is 0.87-1.1 of the fused kernel.  Ragged row tiles (F / dil not a multiple of 4) pad the GEMM: up to 4/3.
extern "C" int aid_conv2d_wino2d_wanted(int B, int Cin, int Cout, int F, int T, int dilF) {
    if (!w2d_shape_ok(Cin, Cout, F, T, dilF)) return 0;
    const W2dGeo ge = w2d_geo(B, F, T, dilF);
    const double pad = (double)ge.J * 4.0 / (double)ge.R;
    if (Cin >= 256 && Cout >= 256) return pad <= 1.34 ? 1 : 0;
    return 0;
}
extern "C" int64_t aid_conv2d_wino2d_positions(int B, int F, int T, int dilF) { return (dilF >= 1 && F % dilF == 0 && T % 4 == 0) ? w2d_geo(B, F, T, dilF).N : 0; }
// per-(sample, group) partial count of stat_ws / dot_ws for x_wino = 3
int aid_w2d_partials(int Cout, int F, int T, int dilF) { return (Cout >> 3) * w2d_nblk(w2d_geo(1, F, T, dilF).NB); }

// aid_conv2d with x_wino = 3: x.p = V [48][Cin][N], wp_wino = U [48][Cin_pad][Cout_pad], ws = scratch for M [48][Cout][N]
int aid_conv53_wino2d(const aid_conv2d_params* p, hipStream_t st) {
    AID_REQUIRE(p->KH == 5 && p->KW == 3 && p->wp_wino && p->wino_taps == 48, "aid_conv2d(x_wino=3): needs the 48-plane pack (aid_pack_conv_weight wpw2)");
    AID_REQUIRE(w2d_shape_ok(p->Cin, p->Cout, p->F, p->T, p->dilF), "aid_conv2d(x_wino=3): shape not supported (aid_conv2d_wino2d_supported)");
    AID_REQUIRE(!p->in_scale && p->act == 0 && !p->x2.p && !p->fin_mode, "aid_conv2d(x_wino=3): no in-kernel prologue, no x2, no fin_mode");
    AID_REQUIRE(!(p->stat_ws && p->dot_ws), "aid_conv2d(x_wino=3): stat_ws and dot_ws are exclusive");
    AID_REQUIRE(!p->dot_ws || p->epi == 1, "aid_conv2d(x_wino=3): dot_ws goes with the dGELU epilogue");
    const W2dGeo ge = w2d_geo(p->B, p->F, p->T, p->dilF);
    AID_REQUIRE(p->ws && p->ws_bytes >= (int64_t)48 * p->Cout * ge.N * 4 && (((uintptr_t)p->ws) & 15) == 0, "aid_conv2d(x_wino=3): ws must hold 48 * Cout * N floats");
    AID_REQUIRE((p->y.sB % 4) == 0 && (p->y.sC % 4) == 0 && (p->y.sF % 4) == 0 && (((uintptr_t)p->y.p) & 15) == 0, "aid_conv2d(x_wino=3): 16-byte aligned output rows");
    AID_REQUIRE(!p->res.p || ((p->res.sB % 4) == 0 && (p->res.sC % 4) == 0 && (p->res.sF % 4) == 0 && (((uintptr_t)p->res.p) & 15) == 0), "aid_conv2d(x_wino=3): 16-byte aligned residual rows");
    AID_REQUIRE(!p->aux.p || ((p->aux.sB % 4) == 0 && (p->aux.sC % 4) == 0 && (p->aux.sF % 4) == 0 && (((uintptr_t)p->aux.p) & 15) == 0), "aid_conv2d(x_wino=3): 16-byte aligned aux rows");
    const int npart = aid_w2d_partials(p->Cout, p->F, p->T, p->dilF);
    AID_REQUIRE(!p->stat_ws || p->stat_n == npart, "aid_conv2d(x_wino=3): stat_n != aid_conv2d_stat_partials()");
    AID_REQUIRE(!p->dot_ws || p->dot_n == npart, "aid_conv2d(x_wino=3): dot_n != aid_conv2d_dot_partials()");
    aid_wino2d_gemm_params gp;
    gp.U = p->wp_wino; gp.V = p->x.p; gp.M = p->ws;
    gp.nxi = 48; gp.Cin = p->Cin; gp.Cout = p->Cout; gp.Cin_pad = p->Cin_pad; gp.Cout_pad = p->Cout_pad; gp.N = ge.N; gp.variant = 0;
    const int rc = aid_wino2d_gemm(&gp, st);
    if (rc != AID_OK) return rc;
    W2dOutDev a;
    a.p = *p; a.M = p->ws;
    a.R = ge.R; a.J = ge.J; a.TG = ge.TG; a.NB = ge.NB; a.N = ge.N; a.nblk = w2d_nblk(ge.NB);
    hipLaunchKernelGGL(w2d_output_kernel, dim3((unsigned)((int64_t)p->B * p->Cout * a.nblk)), dim3(128), 0, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}
