// The 5x3 dilated dense convolution in Winograd form along T on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Once a direct-form kernel (aid_conv_dma.hip) keeps the fp32 matrix pipe busy at the sustained clock, the only way to
// go faster in exact-fp32 arithmetic is to issue FEWER MFMAs.  F(4,3) computes four neighbouring outputs of the 3-tap
// (kw) correlation from 6 products instead of 12; F(8,3) (round 4, points {0, +-0.4, +-0.8, +-1.25, +-2.5, inf}: aid_wino8.h)
// eight outputs from 10 instead of 24.  GEMM view per transform index xi: M = Cout, N = groups of 4 / 8 output samples,
// K = (ci, kh); weights are pre-packed as 30 / 50 "taps" xi*5+kh (aid_pack_conv_weight).
//   conv53_wino4_kernel  : F(4,3), input transform applied when the B fragment is formed (plain activations in LDS)
//   conv53_wino4v_kernel : F(4,3), Winograd-domain input written by the producer pass (aid_scale_act wino=1), 64|96 x 512 tiles (fallback geometries)
//   conv53_wino4r_kernel : F(4,3), Winograd-domain input, ROW-SHARED staging on dilation sub-lattices, 2-3 workgroups per CU (+ pair and split-K instances)
//   conv53_wino8r_kernel : F(8,3) on the same row-shared body (wino4r_tile_body<..., WM = 8>): 64 x 512 tiles, 160 accumulators per lane -- the dominant kernel
//   conv53_wino8r_ks_kernel : its K-group instance for launches of at most one tile per CU (small batches): eight waves on one tile, two K groups, sum through LDS
// Which form takes a launch: wino_form_choice() / aid_conv2d_wino_form().
#include "aid_common.h"
#include "aid_wino8.h"
#include "aid_fin.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvWinoDev {
    aid_conv2d_params p;
    const float* zero;
    int tt_log2, rows_log2, tiles_t, nrows, nchunks;
    int nx, ny, per_xcd;      // F(4,3) kernel: 1-D grid of 8*per_xcd workgroups, swizzled per XCD (see the kernel); per_xcd = 0: plain 2-D grid
};

__device__ float4 g_aid_zero_page_w[16];   // (device symbols are per translation unit without -fgpu-rdc)

#define GLDS16W(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// Direct-to-LDS loads issued through inline asm (used by the triple-buffered Winograd-domain-input kernel): the compiler's
// waitcnt insertion cannot tell which LDS buffer a global_load_lds writes and would drain vmcnt before the next ds_read of
// ANY buffer; issued this way it does not see them and the kernel places its own s_waitcnt vmcnt(n).  vmcnt completes in
// order, so every wait the compiler emits for its own loads stays correct (it can only over-wait).  M0 is a reserved
// register: nothing else in these kernels keeps a value in it.
#define AID_LDS_ADDR(lptr) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(lptr))
#define AID_DMA16_RAW(gptr, lds_addr) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gptr), "s"(lds_addr) : "memory")
// the same with a SCALAR 64-bit base and a 32-bit per-lane byte offset (one VGPR per piece instead of a 64-bit pointer and a per-lane stride)
#define AID_DMA16_SBASE(voff, sbase, lds_addr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory")
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#define AID_VMCNT(n) ((((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)))
#define AID_LGKMCNT0 (15 | (7 << 4) | (0 << 8) | (3 << 14))

template <typename F, int... I>
__device__ __forceinline__ void aid_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void aid_static_for(F&& f) { aid_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// s_waitcnt vmcnt(n) for a wave-uniform n in [0, NMAX] (the instruction takes an immediate)
template <int NMAX>
__device__ __forceinline__ void aid_wait_vmcnt_le(int n) {
    if constexpr (NMAX == 0) { __builtin_amdgcn_s_waitcnt(AID_VMCNT(0)); }
    else { if (n >= NMAX) __builtin_amdgcn_s_waitcnt(AID_VMCNT(NMAX)); else aid_wait_vmcnt_le<NMAX - 1>(n); }
}

// ---- F(4,3): 6 products per 4 outputs (2x fewer MFMAs than direct).  tau now indexes GROUPS of 4 output samples:
//   V = B^T d (d = 6 samples 4g-1 .. 4g+4):  V0 = 4d0-5d2+d4, V1 = (d3+d4)-4(d1+d2), V2 = (d4-d3)+4(d1-d2),
//                                             V3 = (d4-d2)+2(d3-d1), V4 = (d4-d2)-2(d3-d1), V5 = 4d1-5d3+d5
//   U = G w: w0/4, -(w0+w1+w2)/6, -(w0-w1+w2)/6, (w0+2w1+4w2)/24, (w0-2w1+4w2)/24, w2   (30 "taps" xi*5+kh)
//   y0 = M0+M1+M2+M3+M4, y1 = (M1-M2)+2(M3-M4), y2 = (M1+M2)+4(M3+M4), y3 = (M1-M2)+8(M3-M4)+M5
// fp32 error of F(4,3) measured at 2e-6 relative for K=1280 (direct fp32 accumulation: 1e-6).
// MT m-tiles x NTT group-tiles (32 groups = 128 outputs) per wave; N_BLK = 128*NTT*WGN outputs
template <int MT, int NTT, int WGM, int WGN, int RMAX, int KC, int MINW>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv53_wino4_kernel(const ConvWinoDev a) {
    constexpr int KH = 5, NXI = 6, TAPS = NXI * KH;     // 30 transformed taps
    constexpr int NW = WGM * WGN;
    constexpr int NTHREADS = 64 * NW;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 128 * NTT * WGN;              // output samples per tile
    constexpr int HALO = 2 * RMAX;
    constexpr int XB = N_BLK + HALO;
    constexpr int XSZ = KC * KH * XB;
    constexpr int WROW = (M_BLK % 64 == 0) ? M_BLK : ((M_BLK + 63) / 64) * 64;
    constexpr int WSZ_RAW = TAPS * KC * WROW;
    constexpr int WSZ = ((WSZ_RAW + 255) / 256) * 256;
    constexpr int BUFSZ = XSZ + WSZ;
    constexpr int NXP = KC * KH * (N_BLK / 256);
    constexpr int NWP = WSZ / 256;
    constexpr int NP = NXP + NWP;
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int HQ = (2 * KC * KH * RMAX + NTHREADS - 1) / NTHREADS;
    constexpr int NSTEP = KH * (KC / 2);                // (kh, ci-pair) steps per chunk, 4*MT*NTT MFMAs each
    static_assert(N_BLK % 256 == 0, "x tile must be whole 1-KiB pieces");

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;

    // Two STATIC buffers (not one dynamic array): distinct LDS objects carry alias scopes after LDS lowering, so the
    // waitcnt pass knows the ds_reads of chunk c cannot touch the buffer the direct-to-LDS loads of chunk c+1 write,
    // and leaves those loads in flight for the whole chunk (with a single dynamic array it emitted s_waitcnt vmcnt(0)
    // right after issuing them: the "asynchronous" staging was synchronous).
    __shared__ __attribute__((aligned(16))) float sbuf0[BUFSZ];
    __shared__ __attribute__((aligned(16))) float sbuf1[BUFSZ];
    __shared__ int rowinfo[2 * RMAX];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN;
    const int wn = wave % WGN;

    // XCD-aware tile order.  Hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2).
    // Logical tile L = xcd * per_xcd + (id / 8): every XCD walks ONE contiguous range of (row-group, m-tile) pairs, m-tile
    // fastest, so the Cout tiles of the same activations and the row groups that share dilated rows meet in the same L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.per_xcd > 0) {
        const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
        if (Lt >= a.nx * a.ny) return;
        bx = Lt / a.ny;
        by = Lt - bx * a.ny;
    }
    const int tile_t = bx % a.tiles_t;
    const int rg = bx / a.tiles_t;
    const int row0 = rg << a.rows_log2;
    const int t0 = tile_t << a.tt_log2;
    const int m0 = by * M_BLK;

    for (int r = tid; r < ROWS; r += NTHREADS) {
        const int rid = row0 + r;
        int b = -1, f = 0;
        if (rid < a.nrows) { b = rid / p.F; f = rid - b * p.F; }
        rowinfo[2 * r] = b;
        rowinfo[2 * r + 1] = f;
    }
    __syncthreads();

    // ---- DMA piece descriptors (identical scheme to conv53_dma_kernel; weights use the 20-tap Winograd pack) ----
    const float* psrc[PPW];
    int pstride[PPW], plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        psrc[i] = a.zero; pstride[i] = 0; plds[i] = -1;
        if (pc < NXP) {
            const int blk = pc / (N_BLK / 256), sub = pc % (N_BLK / 256);
            const int ci = blk / KH, kh = blk % KH;
            const int n = sub * 256 + 4 * lane;
            const int rr = n >> a.tt_log2, tt = n & (TT - 1);
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            plds[i] = blk * XB + sub * 256;
            if (b >= 0 && fi >= 0 && fi < p.F && t0 + tt < p.T) {
                psrc[i] = p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF + t0 + tt;
                pstride[i] = (int)(KC * p.x.sC);
            }
        } else if (pc < NP) {
            const int wp_ = pc - NXP;
            const int e = wp_ * 256 + 4 * lane;
            const int row = e / WROW, col = e % WROW;
            const int tap = row / KC, ci = row % KC;
            plds[i] = XSZ + wp_ * 256;
            if (col < M_BLK && e < WSZ_RAW) {
                psrc[i] = p.wp_wino + ((int64_t)tap * p.Cin_pad + ci) * p.Cout_pad + m0 + col;
                pstride[i] = KC * p.Cout_pad;
            }
        }
    }
    const int nhalo = 2 * KC * KH * ROWS;
    const float* hsrc[HQ];
    int hstride[HQ], hlds[HQ];
#pragma unroll
    for (int i = 0; i < HQ; ++i) {
        const int h = tid + i * NTHREADS;
        hsrc[i] = a.zero; hstride[i] = 0; hlds[i] = -1;
        if (h < nhalo) {
            const int side = h & 1;
            const int rr = (h >> 1) & (ROWS - 1);
            const int blk = (h >> 1) >> a.rows_log2;
            const int ci = blk / KH, kh = blk % KH;
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            const int t = side ? (t0 + TT) : (t0 - 1);
            hlds[i] = blk * XB + N_BLK + rr * 2 + side;
            if (b >= 0 && fi >= 0 && fi < p.F && t >= 0 && t < p.T) {
                hsrc[i] = p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF + t;
                hstride[i] = (int)(KC * p.x.sC);
            }
        }
    }
    // ---- operand addresses -----------------------------------------------------------------------------------------
    const int half = lane >> 5;
    const int ttau_log2 = a.tt_log2 - 2;                 // groups per row = TT/4
    int vD0[NTT], vD12[NTT], vD3[NTT];                   // float offsets of d0 | (d1..d4) | d5 inside a (ci,kh) block
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const int nt = (wn * NTT + j) * 32 + (lane & 31);            // tau index inside the tile
        const int rr = nt >> ttau_log2, tau = nt & ((TT >> 2) - 1);
        const int n = rr * TT + 4 * tau;                             // core position of d1
        vD12[j] = half * KH * XB + n;
        vD0[j] = half * KH * XB + ((tau == 0) ? (N_BLK + rr * 2) : (n - 1));
        vD3[j] = half * KH * XB + ((4 * tau + 4 >= TT) ? (N_BLK + rr * 2 + 1) : (n + 4));
    }
    int vA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) vA[i] = XSZ + half * WROW + (wm * MT + i) * 32 + (lane & 31);

    f32x16 acc[MT][NTT][NXI];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int x = 0; x < NXI; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][x][r] = 0.f;

    float hv[HQ];
    auto issue_dma = [&](int ch, float* buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (plds[i] >= 0) {
                const float* src = psrc[i] + (int64_t)ch * pstride[i];
                GLDS16W(src, buf + plds[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < HQ; ++i) hv[i] = hsrc[i][(int64_t)ch * hstride[i]];
    };
    auto write_halo = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < HQ; ++i)
            if (hlds[i] >= 0) buf[hlds[i]] = hv[i];
    };

    issue_dma(0, sbuf0);
    write_halo(sbuf0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = cur ? sbuf1 : sbuf0;
        float* Nx = cur ? sbuf0 : sbuf1;
        const bool more = (ch + 1) < a.nchunks;
        if (more) issue_dma(ch + 1, Nx);
        // raw samples of step s+1 are fetched while the MFMAs of step s run
        float d0[2][NTT], d3[2][NTT];
        float4 d12[2][NTT];
        float av[2][MT][NXI];
        auto load_step = [&](int s, int buf) {
            const int kh = s / (KC / 2), cp = s % (KC / 2);
            const int xo = (2 * cp * KH + kh) * XB;
#pragma unroll
            for (int j = 0; j < NTT; ++j) {
                d0[buf][j] = Bf[vD0[j] + xo];
                d12[buf][j] = *reinterpret_cast<const float4*>(Bf + vD12[j] + xo);
                d3[buf][j] = Bf[vD3[j] + xo];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int x = 0; x < NXI; ++x)
                    av[buf][i][x] = Bf[vA[i] + ((x * KH + kh) * KC + 2 * cp) * WROW];
        };
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s == 0) __builtin_amdgcn_s_setprio(3);
            else if (s == (NSTEP + 3) / 4) __builtin_amdgcn_s_setprio(2);
            else if (s == (NSTEP + 1) / 2) __builtin_amdgcn_s_setprio(1);
            else if (s == (3 * NSTEP + 3) / 4) __builtin_amdgcn_s_setprio(0);
            if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
            const int bq = s & 1;
#pragma unroll
            for (int j = 0; j < NTT; ++j) {
                const float e0 = d0[bq][j], e1 = d12[bq][j].x, e2 = d12[bq][j].y, e3 = d12[bq][j].z, e4 = d12[bq][j].w, e5 = d3[bq][j];
                const float s12 = e1 + e2, m12 = e1 - e2, m42 = e4 - e2, m31 = e3 - e1;
                const float V[NXI] = {4.f * e0 - 5.f * e2 + e4, (e3 + e4) - 4.f * s12, (e4 - e3) + 4.f * m12,
                                      m42 + 2.f * m31, m42 - 2.f * m31, 4.f * e1 - 5.f * e3 + e5};
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int x = 0; x < NXI; ++x)
                        acc[i][j][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bq][i][x], V[x], acc[i][j][x], 0, 0, 0);
            }
        }
        if (more) write_halo(Nx);
        __syncthreads();
    };
    for (int ch = 0; ch < a.nchunks; ch += 2) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < a.nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: output transform, then gate / dGELU / residual, four samples per lane -------------------------------
    float dsum[MT][4];                                   // <y, aux> per block of 4 rows (dot_ws)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) dsum[i][q] = 0.f;
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const int nt = (wn * NTT + j) * 32 + (lane & 31);
        const int rr = nt >> ttau_log2, tau = nt & ((TT >> 2) - 1);
        const int b = rowinfo[2 * rr];
        const int f = rowinfo[2 * rr + 1];
        const int t = t0 + 4 * tau;
        if (b < 0 || t >= p.T) continue;                 // T % 4 == 0: the four samples are in range together
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 4) {         // batches of 4 rows: gather, then compute + store
                float4 rv[4], ur[4];                      // residual, raw aux (scaled by `as` when the dGELU is evaluated)
                float sv[4], as[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + q;
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    const bool ok = m < p.Cout;
                    rv[q] = (ok && p.res.p) ? *reinterpret_cast<const float4*>(p.res.p + rbase + (int64_t)m * p.res.sC) : make_float4(0.f, 0.f, 0.f, 0.f);
                    sv[q] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                    if (ok && p.epi == 1) {
                        as[q] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                        ur[q] = *reinterpret_cast<const float4*>(p.aux.p + abase + (int64_t)m * p.aux.sC);
                    } else { as[q] = 0.f; ur[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + q;
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m >= p.Cout) continue;
                    const float M0 = acc[i][j][0][r], M1 = acc[i][j][1][r], M2 = acc[i][j][2][r], M3 = acc[i][j][3][r],
                                M4 = acc[i][j][4][r], M5 = acc[i][j][5][r];
                    const float a12 = M1 + M2, s12 = M1 - M2, a34 = M3 + M4, s34 = M3 - M4;
                    float y0 = (M0 + a12 + a34) * sv[q];
                    float y1 = (s12 + 2.f * s34) * sv[q];
                    float y2 = (a12 + 4.f * a34) * sv[q];
                    float y3 = (s12 + 8.f * s34 + M5) * sv[q];
                    if (p.epi == 1) { y0 *= aid_dgelu(ur[q].x * as[q]); y1 *= aid_dgelu(ur[q].y * as[q]); y2 *= aid_dgelu(ur[q].z * as[q]); y3 *= aid_dgelu(ur[q].w * as[q]); }
                    y0 += p.res_scale * rv[q].x; y1 += p.res_scale * rv[q].y; y2 += p.res_scale * rv[q].z; y3 += p.res_scale * rv[q].w;
                    y0 *= p.alpha; y1 *= p.alpha; y2 *= p.alpha; y3 *= p.alpha;
                    *reinterpret_cast<float4*>(p.y.p + ybase + (int64_t)m * p.y.sC) = make_float4(y0, y1, y2, y3);
                    if (p.dot_ws) dsum[i][r0 >> 2] += (y0 * ur[q].x + y1 * ur[q].y) + (y2 * ur[q].z + y3 * ur[q].w);
                }
            }
        }
    }
    // ---- optional: <y, aux> per (sample, channel group), one partial per tile (replaces the aid_group_dot pass) -----------
    if (p.dot_ws) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = dsum[i][q];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);     // the 32 lanes of a half-wave share their rows
                dsum[i][q] = v;
            }
        float* red = sbuf0;                              // (every wave is past the last chunk's barrier)
        if ((lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((wave * 2 + half) * MT + i) * 4 + q] = dsum[i][q];
        }
        __syncthreads();
        const int cpg = p.Cout >> 3;                     // channels per group (8 groups); M_BLK % cpg == 0 (host-checked)
        const int g = m0 / cpg + tid;
        if (tid < M_BLK / cpg && g < 8) {
            float sacc = 0.f;
            for (int w = 0; w < NW; ++w)                 // fixed order: deterministic
                for (int h = 0; h < 2; ++h)
                    for (int i = 0; i < MT; ++i)
                        for (int q = 0; q < 4; ++q) {
                            const int mrow = m0 + ((w / WGN) * MT + i) * 32 + 4 * h + 8 * q;
                            if (mrow / cpg == g) sacc += red[((w * 2 + h) * MT + i) * 4 + q];
                        }
            const int b = rowinfo[0];
            const int ptile = ((row0 - b * p.F) >> a.rows_log2) * a.tiles_t + tile_t;
            p.dot_ws[((int64_t)b * 8 + g) * p.dot_n + ptile] = (double)sacc;
        }
    }
}

// ---- F(4,3) with the input transform done by the producer (aid_scale_act(wino=1), x_wino = 1) -----------------------------
// x is V = B^T d, [B, Cin, F, 6, T/4].  The kernel only stages and multiplies: no halo samples, no transform VALU at
// fragment-load time, and every B fragment is a conflict-free ds_read_b32 of 32 consecutive groups of one plane
// (the in-kernel transform read d0 / d5 with stride 4: 4-way bank conflicts on a third of the LDS reads).
// LDS block per (ci,kh): six planes [xi][N_BLK/4 groups]; a 1-KiB direct-to-LDS piece = two half-planes.
template <int MT, int NTT, int WGM, int WGN, int RMAX, int KC, int MINW>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv53_wino4v_kernel(const ConvWinoDev a) {
    constexpr int KH = 5, NXI = 6, TAPS = NXI * KH;     // 30 transformed taps
    constexpr int NW = WGM * WGN;
    constexpr int NTHREADS = 64 * NW;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 128 * NTT * WGN;              // output samples per tile
    constexpr int NG = N_BLK / 4;                       // groups of 4 outputs per tile = length of one V plane in LDS
    constexpr int XB = NXI * NG;                        // a (ci,kh) block: six planes [xi][group]
    constexpr int XSZ = KC * KH * XB;
    constexpr int WROW = M_BLK;                         // (96-wide tiles: no padding, 3 x 54 KB of LDS)
    constexpr int WSZ_RAW = TAPS * KC * WROW;
    constexpr int WSZ = ((WSZ_RAW + 255) / 256) * 256;
    constexpr int BUFSZ = XSZ + WSZ;
    constexpr int NXP = XSZ / 256;                       // 1-KiB pieces over the whole x region (a piece may straddle two (ci,kh) blocks)
    constexpr int NWP = WSZ / 256;
    constexpr int NP = NXP + NWP;
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int NSTEP = KH * (KC / 2);                // (kh, ci-pair) steps per chunk, 4*MT*NTT MFMAs each
    static_assert(XSZ % 256 == 0 && NG % 4 == 0, "whole 1-KiB pieces; a lane's 4 groups stay inside one plane");

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;

    // THREE static buffers and a SPREAD issue: the direct-to-LDS loads of chunk c+2 are issued one or two per k-step
    // while chunk c is multiplied.  Issuing a chunk's ~45 KiB of loads back to back at the top of a chunk (8 waves x 6
    // instructions, 16 texture-address cycles each) back-pressured the vector-memory issue and held every wave's MFMA stream
    // behind it: removing the loads gave +15-22 % (PMC: matrix pipe 76 -> 88 % busy at the same clock), a deeper prefetch
    // alone nothing.  Spreading needs the extra chunk of distance: the last pieces are issued near the end of a chunk.
    __shared__ __attribute__((aligned(16))) float sbuf0[BUFSZ];
    __shared__ __attribute__((aligned(16))) float sbuf1[BUFSZ];
    __shared__ __attribute__((aligned(16))) float sbuf2[BUFSZ];
    __shared__ int rowinfo[2 * RMAX];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN;
    const int wn = wave % WGN;

    // XCD-aware tile order.  Hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2).
    // Logical tile L = xcd * per_xcd + (id / 8): every XCD walks ONE contiguous range of (row-group, m-tile) pairs, m-tile
    // fastest, so the Cout tiles of the same activations and the row groups that share dilated rows meet in the same L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.per_xcd > 0) {
        const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
        if (Lt >= a.nx * a.ny) return;
        bx = Lt / a.ny;
        by = Lt - bx * a.ny;
    }
    const int tile_t = bx % a.tiles_t;
    const int rg = bx / a.tiles_t;
    const int row0 = rg << a.rows_log2;
    const int t0 = tile_t << a.tt_log2;
    const int m0 = by * M_BLK;

    for (int r = tid; r < ROWS; r += NTHREADS) {
        const int rid = row0 + r;
        int b = -1, f = 0;
        if (rid < a.nrows) { b = rid / p.F; f = rid - b * p.F; }
        rowinfo[2 * r] = b;
        rowinfo[2 * r + 1] = f;
    }
    __syncthreads();

    // ---- DMA piece descriptors (identical scheme to conv53_dma_kernel; weights use the 20-tap Winograd pack) ----
    const float* psrc[PPW];
    int pstride[PPW], plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        psrc[i] = a.zero; pstride[i] = 0; plds[i] = -1;
        if (pc < NXP) {
            const int eg = pc * 256 + 4 * lane;          // float offset inside the x region = blk * XB + xi * NG + group
            const int blk = eg / XB, e = eg % XB;
            const int ci = blk / KH, kh = blk % KH;
            const int xi = e / NG, gq = e % NG;
            const int rr = gq >> (a.tt_log2 - 2), gl = gq & ((TT >> 2) - 1);
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            plds[i] = pc * 256;
            if (b >= 0 && fi >= 0 && fi < p.F && t0 + 4 * gl < p.T) {
                psrc[i] = p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF + (int64_t)xi * (p.T >> 2) + (t0 >> 2) + gl;
                pstride[i] = (int)(KC * p.x.sC);
            }
        } else if (pc < NP) {
            const int wp_ = pc - NXP;
            const int e = wp_ * 256 + 4 * lane;
            const int row = e / WROW, col = e % WROW;
            const int tap = row / KC, ci = row % KC;
            plds[i] = XSZ + wp_ * 256;
            if (col < M_BLK && e < WSZ_RAW) {
                psrc[i] = p.wp_wino + ((int64_t)tap * p.Cin_pad + ci) * p.Cout_pad + m0 + col;
                pstride[i] = KC * p.Cout_pad;
            }
        }
    }
    // ---- operand addresses -----------------------------------------------------------------------------------------
    const int half = lane >> 5;
    const int ttau_log2 = a.tt_log2 - 2;                 // groups per row = TT/4
    int vB[NTT];                                         // float offset of V[0][group] inside a (ci,kh) block
#pragma unroll
    for (int j = 0; j < NTT; ++j) vB[j] = half * KH * XB + (wn * NTT + j) * 32 + (lane & 31);
    int vA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) vA[i] = XSZ + half * WROW + (wm * MT + i) * 32 + (lane & 31);

    f32x16 acc[MT][NTT][NXI];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int x = 0; x < NXI; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][x][r] = 0.f;

    static_assert(NP > (PPW - 1) * NW, "every wave owns PPW or PPW-1 pieces");
    const int mine = (wave + (PPW - 1) * NW < NP) ? PPW : PPW - 1;      // loads this wave issues per chunk (exact)
    auto issue_piece = [&](auto ic, int ch, float* buf) {
        constexpr int i = decltype(ic)::value;
        if (wave + i * NW < NP) {
            const float* src = psrc[i] + (int64_t)ch * pstride[i];
            const unsigned la = AID_LDS_ADDR(buf + plds[i]);
            AID_DMA16_RAW(src, la);
        }
    };
    auto issue_all = [&](int ch, float* buf) { aid_static_for<PPW>([&](auto ic) { issue_piece(ic, ch, buf); }); };
    // pieces i with i % NSTEP == s are issued at k-step s
    auto issue_step = [&](auto sc, int ch, float* buf) {
        aid_static_for<PPW>([&](auto ic) {
            if constexpr (decltype(ic)::value % NSTEP == decltype(sc)::value) issue_piece(ic, ch, buf);
        });
    };

    issue_all(0, sbuf0);
    if (a.nchunks > 1) issue_all(1, sbuf1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = cur == 0 ? sbuf0 : (cur == 1 ? sbuf1 : sbuf2);
        float* Nx = cur == 0 ? sbuf2 : (cur == 1 ? sbuf0 : sbuf1);       // buffer (cur + 2) % 3: read during the previous chunk
        const bool more = (ch + 2) < a.nchunks;
        // fragments of step s+1 are fetched while the MFMAs of step s run
        float bv[2][NTT][NXI];
        float av[2][MT][NXI];
        auto load_step = [&](int s, int buf) {
            const int kh = s / (KC / 2), cp = s % (KC / 2);
            const int xo = (2 * cp * KH + kh) * XB;
#pragma unroll
            for (int j = 0; j < NTT; ++j)
#pragma unroll
                for (int x = 0; x < NXI; ++x) bv[buf][j][x] = Bf[vB[j] + xo + x * NG];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int x = 0; x < NXI; ++x)
                    av[buf][i][x] = Bf[vA[i] + ((x * KH + kh) * KC + 2 * cp) * WROW];
        };
        load_step(0, 0);
        aid_static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s == 0) __builtin_amdgcn_s_setprio(3);
            else if (s == (NSTEP + 3) / 4) __builtin_amdgcn_s_setprio(2);
            else if (s == (NSTEP + 1) / 2) __builtin_amdgcn_s_setprio(1);
            else if (s == (3 * NSTEP + 3) / 4) __builtin_amdgcn_s_setprio(0);
            if (more) issue_step(sc, ch + 2, Nx);        // this step's share of the loads of chunk ch+2
            if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
            constexpr int bq = s & 1;
#pragma unroll
            for (int j = 0; j < NTT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int x = 0; x < NXI; ++x)
                        acc[i][j][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bq][i][x], bv[bq][j][x], acc[i][j][x], 0, 0, 0);
        });
        // End of chunk: the loads of chunk ch+1 (issued during the previous chunk) must have landed; those of chunk ch+2
        // (`mine` instructions of this wave, issued during this chunk) stay in flight.  __syncthreads() would drain them
        // (its release fence waits for every outstanding VMEM store, LDS-DMA included), so the barrier is spelled out.
        asm volatile("" ::: "memory");
        if (more) {
            if (mine == PPW) __builtin_amdgcn_s_waitcnt(AID_VMCNT(PPW)); else __builtin_amdgcn_s_waitcnt(AID_VMCNT(PPW - 1));
        } else {
            __builtin_amdgcn_s_waitcnt(AID_VMCNT(0));
        }
        __builtin_amdgcn_s_waitcnt(AID_LGKMCNT0);        // this wave's reads of the buffer that is overwritten next
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int ch = 0; ch < a.nchunks; ch += 3) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < a.nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
        if (ch + 2 < a.nchunks) chunk(std::integral_constant<int, 2>{}, ch + 2);
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: output transform, then gate / dGELU / residual, four samples per lane -------------------------------
    float dsum[MT][4];                                   // <y, aux> per block of 4 rows (dot_ws)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) dsum[i][q] = 0.f;
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const int nt = (wn * NTT + j) * 32 + (lane & 31);
        const int rr = nt >> ttau_log2, tau = nt & ((TT >> 2) - 1);
        const int b = rowinfo[2 * rr];
        const int f = rowinfo[2 * rr + 1];
        const int t = t0 + 4 * tau;
        if (b < 0 || t >= p.T) continue;                 // T % 4 == 0: the four samples are in range together
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 4) {         // batches of 4 rows: gather, then compute + store
                float4 rv[4], ur[4];                      // residual, raw aux (scaled by `as` when the dGELU is evaluated)
                float sv[4], as[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + q;
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    const bool ok = m < p.Cout;
                    rv[q] = (ok && p.res.p) ? *reinterpret_cast<const float4*>(p.res.p + rbase + (int64_t)m * p.res.sC) : make_float4(0.f, 0.f, 0.f, 0.f);
                    sv[q] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                    if (ok && p.epi == 1) {
                        as[q] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                        ur[q] = *reinterpret_cast<const float4*>(p.aux.p + abase + (int64_t)m * p.aux.sC);
                    } else { as[q] = 0.f; ur[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + q;
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m >= p.Cout) continue;
                    const float M0 = acc[i][j][0][r], M1 = acc[i][j][1][r], M2 = acc[i][j][2][r], M3 = acc[i][j][3][r],
                                M4 = acc[i][j][4][r], M5 = acc[i][j][5][r];
                    const float a12 = M1 + M2, s12 = M1 - M2, a34 = M3 + M4, s34 = M3 - M4;
                    float y0 = (M0 + a12 + a34) * sv[q];
                    float y1 = (s12 + 2.f * s34) * sv[q];
                    float y2 = (a12 + 4.f * a34) * sv[q];
                    float y3 = (s12 + 8.f * s34 + M5) * sv[q];
                    if (p.epi == 1) { y0 *= aid_dgelu(ur[q].x * as[q]); y1 *= aid_dgelu(ur[q].y * as[q]); y2 *= aid_dgelu(ur[q].z * as[q]); y3 *= aid_dgelu(ur[q].w * as[q]); }
                    y0 += p.res_scale * rv[q].x; y1 += p.res_scale * rv[q].y; y2 += p.res_scale * rv[q].z; y3 += p.res_scale * rv[q].w;
                    y0 *= p.alpha; y1 *= p.alpha; y2 *= p.alpha; y3 *= p.alpha;
                    *reinterpret_cast<float4*>(p.y.p + ybase + (int64_t)m * p.y.sC) = make_float4(y0, y1, y2, y3);
                    if (p.dot_ws) dsum[i][r0 >> 2] += (y0 * ur[q].x + y1 * ur[q].y) + (y2 * ur[q].z + y3 * ur[q].w);
                }
            }
        }
    }
    // ---- optional: <y, aux> per (sample, channel group), one partial per tile (replaces the aid_group_dot pass) -----------
    if (p.dot_ws) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = dsum[i][q];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);     // the 32 lanes of a half-wave share their rows
                dsum[i][q] = v;
            }
        float* red = sbuf0;                              // (every wave is past the last chunk's barrier)
        if ((lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((wave * 2 + half) * MT + i) * 4 + q] = dsum[i][q];
        }
        __syncthreads();
        const int cpg = p.Cout >> 3;                     // channels per group (8 groups); M_BLK % cpg == 0 (host-checked)
        const int g = m0 / cpg + tid;
        if (tid < M_BLK / cpg && g < 8) {
            float sacc = 0.f;
            for (int w = 0; w < NW; ++w)                 // fixed order: deterministic
                for (int h = 0; h < 2; ++h)
                    for (int i = 0; i < MT; ++i)
                        for (int q = 0; q < 4; ++q) {
                            const int mrow = m0 + ((w / WGN) * MT + i) * 32 + 4 * h + 8 * q;
                            if (mrow / cpg == g) sacc += red[((w * 2 + h) * MT + i) * 4 + q];
                        }
            const int b = rowinfo[0];
            const int ptile = ((row0 - b * p.F) >> a.rows_log2) * a.tiles_t + tile_t;
            p.dot_ws[((int64_t)b * 8 + g) * p.dot_n + ptile] = (double)sacc;
        }
    }
}

// ---- F(4,3), Winograd-domain input, ROW-SHARED staging, two workgroups per CU ---------------------------------------------------
// A tile is 64 output channels x 256 positions = RA rows of ONE dilation sub-lattice (f, f+d, f+2d, ...) x TT samples, so that the
// five dilated input rows of neighbouring output rows coincide: RA + 4 input rows are staged per input channel instead of 5 RA
// (x staging per chunk: 6 KB instead of 15 KB for this tile size; with the 15 KB of weights: 21 KB per 120 MFMAs = what the 64 x 512
// tile moves per MFMA).  Four waves (one per SIMD) and 3 x 21.5 KB of LDS let TWO workgroups share a CU: their phases drift apart,
// and one workgroup's prologue / epilogue (24 us per tile that the one-workgroup-per-CU kernel exposes, profiles/r02_epilogue_gates.txt)
// runs under the other's K loop.  Layout of the x region per input channel: [slot 0..RA+3][xi 0..5][TT/4 groups]; output row j reads
// slot j + kh.  Requires F % dilF == 0 (rows of a residue class: F/dilF).
// Large dilations leave few rows per residue class (F/dilF = 28, 14, 7 on the deepest level): there a tile takes RA rows of each of NC
// ADJACENT residue classes (NC x RA x TT = 256 positions, NC (RA + 4) staged rows), so that RA divides the class without padding rows.
#define AID_W4R_SPLIT_MAX_TILES 448
#define AID_W4R_SPLIT_FLAG_BYTES AID_CONV2D_SPLIT_FLAG_BYTES
static_assert(AID_W4R_SPLIT_MAX_TILES * 2 * 4 <= AID_CONV2D_SPLIT_FLAG_BYTES - 4, "flag region (its last word is the sticky error word)");
static inline int64_t aid_w4r_split_bytes(int64_t ntiles) { return AID_W4R_SPLIT_FLAG_BYTES + ntiles * (int64_t)(96 * 256 * 4); }

struct W4rGeo {                // one tile family of a launch
    int quads, ttiles, ny, per_xcd, ntiles, rgroups;
    int m_base, m_stride;      // Cout tile `by` starts at m_base + by * m_stride
    int dot_base;              // first per-sample partial slot of this family
};
struct ConvWinoRDev {
    aid_conv2d_params p;
    const float* zero;
    int nchunks;               // chunks of 2 input channels PER SPLIT
    int dot_all;               // 96-channel layers (two tile families): every tile writes all eight groups of its own partial slots
    int splits;                // 1, or 2: the K axis of a tile is shared by two workgroups (see the kernel)
    float* part;               // splits == 2: [tile][96][256] Winograd-domain partial accumulators of the workgroup that finishes first
    unsigned* flags;           // splits == 2: [tile][2] (arrival counter, partial published); zero before and after every launch
    W4rGeo g[2];               // [1]: the 32-channel remainder tiles of a 96-channel layer (pair instances)
    int fin_total;             // fin_mode: tiles per sample (both families) -- the tile that finds fin_count[b] == fin_total - 1 folds the sample's partials
};

// fin_mode (aid_kernels.h): the last tile of sample b folds the epilogue partials of the whole sample -- the work of group_stats_final / norm_bwd_coef
// (aid_norm.hip), same summation order (lane l adds partials l, l + 64, ..., then the xor tree), hence the same bits -- so that those two 5-10 us
// launches per normalisation layer (194 per guided evaluation, a twentieth of a batch-1 evaluation) are not made.  The partials were published with
// agent-scope stores before the arrival counter was bumped (as the split-K exchange above); they are read back past the XCD's L2 the same way.
// Tried for small grids (round 3, profiles/r03_half_tile_probe.txt): two-wave workgroups on half the positions (64 x 128 / 32 x 256 tiles, twice
// the workgroups, four per CU) when a launch has fewer than 300 ... 1100 full tiles -- B = 1: 24.7 -> 29.4 ... 30.2 ms of 5x3 time per guided
// evaluation, B = 2 / 3 neutral to -12 %: the weight slice every tile re-stages is amortised over half the MFMAs.  Not kept.
// WGM = 2: 64 output channels x 256 positions; WGM = 1: 32 output channels x 512 positions (the remainder tile of 96-channel layers,
// whose first 64 channels take the 64-wide tile: four waves per workgroup either way, one per SIMD)
// NB: LDS buffers asked for.  3 (default): two workgroups per CU; 2: 43 KB, THREE workgroups per CU -- measured +1..3 % on the short-K layers
// (Cin <= 128: more of the tile is prologue / epilogue, which a third resident workgroup covers) and -0.5 % on Cin = 256 (profiles/r02_wino4r_probe.txt).
// x region layout per input channel: slots are interleaved in units of IL = 32 / GPR rows -- [slot / IL][xi][slot % IL][GPR groups] -- so that
// the IL adjacent rows one half-wave ds_read_b32 touches (lanes of a half-wave own 32 consecutive groups = IL rows) fall on 32 distinct
// banks for every kh (with the plain [slot][xi][GPR] order rows are 6*GPR floats = 0 mod 32 banks apart: every B read was 2-way conflicted,
// a third of the kernel's LDS cycles, profiles/r02_wino_pmc.txt; now 0, profiles/r03_wino4r_staging_pmc.txt).  Classes with CSLOT % IL != 0
// (NC >= 4) keep the plain order.  Measured with it (round 3, profiles/r03_wino4r_staging_ab.txt): staging the RAW 15 taps and forming U = G w
// in registers (half the weight bytes, 14 instead of 21 direct-to-LDS pieces per chunk, 3 instead of 6 A reads per k-step, 2x the VALU) and the
// conflict-free layout change the kernel time by less than 0.5 % in either direction: it is bound by the matrix pipe (84 % busy at K = 256)
// and by the prologue / epilogue of a tile, not by LDS or staging any more.  The interleave stays (no conflicts, no cost); raw-tap staging does not.
// WM: outputs per Winograd group -- 4: F(4,3), 6 products per group; 8: F(8,3), 10 products per group (aid_wino8.h), tiles of twice the positions
// KS: K groups per workgroup -- 2: the NWV waves form two groups of NWV / 2, every staged chunk holds 2 x KC input channels and group k multiplies
// channels [k KC, (k + 1) KC) of it; the groups add their accumulators through LDS after the K loop (conv53_wino8r_ks_kernel)
template <int TT, int NC, int WGM, int NB, int WPC, int NWV, int WM = 4, int KS = 1>
struct W4rShape {
    static constexpr int KH = 5, NXI = WM + 2, TAPS = NXI * KH, KC = 2, KCS = KC * KS;
    static constexpr int NW = NWV, NWQ = NW / KS, WGN = NWQ / WGM;
    static constexpr int M_BLK = 32 * WGM, N_BLK = 32 * WM * WGN;
    static constexpr int GPR = TT / WM;                        // groups of WM outputs per row
    static constexpr int RA = N_BLK / (TT * NC);               // output rows per residue class and tile
    static constexpr int CSLOT_USED = RA + KH - 1;             // staged input rows per channel and class
    static constexpr int ILR = 32 / GPR;                       // rows read by one half-wave
    static constexpr int IL = (RA % ILR == 0) ? ILR : 1;       // slot interleave unit (1: plain order)
    // slots per class, padded to a multiple of the interleave unit (the padding slots are never loaded): round 4 -- the T = 32 tiles of the F(8,3)
    // kernel (RA = 16, 20 staged rows, 8 rows per half-wave) otherwise fell back to the plain order and 2-way bank conflicts on every B read
    // (SQ_LDS_BANK_CONFLICT 4.6e7 and 72 % matrix-pipe busy against 82 % of the T = 64 tiles, profiles/r04_wino_pmc.txt)
    static constexpr int CSLOT = (CSLOT_USED + IL - 1) / IL * IL;
    static constexpr int NSLOT = NC * CSLOT;
    static constexpr int XCI = NSLOT * NXI * GPR;              // floats per input channel in the x region
    static constexpr int XSZ = ((KCS * XCI + 255) / 256) * 256;
    static constexpr int WROW = M_BLK;
    static constexpr int WSZ_RAW = TAPS * KCS * WROW;          // 64 channels: 3840 floats = 15 pieces; 32: 1920 -> 8 pieces (the last half used)
    static constexpr int WSZ = ((WSZ_RAW + 255) / 256) * 256;
    static constexpr int BUFSZ = XSZ + WSZ;
    static constexpr int NBUF = (NB == 3 && 3 * BUFSZ * 4 * WPC <= 160 * 1024) ? 3 : 2;  // three buffers only while WPC workgroups still fit a CU
    static constexpr int LDS = NBUF * BUFSZ;                   // floats
};

// One tile.  `smem`: W4rShape::LDS floats of LDS (declared by the kernel, so that the two tile families of a pair instance share it);
// `bid`: workgroup index within this tile family's part of the grid.
template <int TT, int NC, int WGM, int NB, int WPC, int NWV, bool SPK, int WM = 4, int KS = 1>
__device__ __forceinline__ void wino4r_tile_body(const ConvWinoRDev& a, const W4rGeo& ge, float* smem, const int bid) {
    using S = W4rShape<TT, NC, WGM, NB, WPC, NWV, WM, KS>;
    constexpr int KH = S::KH, NXI = S::NXI, KC = S::KC, KCS = S::KCS;
    constexpr int NW = S::NW, NWQ = S::NWQ, WGN = S::WGN;
    static_assert(KS == 1 || (KS == 2 && !SPK), "K groups: not combined with the split-K exchange");
    constexpr int M_BLK = S::M_BLK;
    constexpr int GPR = S::GPR, RA = S::RA, CSLOT = S::CSLOT, CSLOT_USED = S::CSLOT_USED, IL = S::IL, XCI = S::XCI, XSZ = S::XSZ, WROW = S::WROW;
    constexpr int WSZ_RAW = S::WSZ_RAW, WSZ = S::WSZ, BUFSZ = S::BUFSZ, NBUF = S::NBUF;
    constexpr int NXP = XSZ / 256, NWP = WSZ / 256, NP = NXP + NWP;
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int NSTEP = KH;
    // steps of a chunk the next chunk's direct-to-LDS loads are issued in.  F(8,3) (two buffers: the loads of chunk c + 1 must have landed when chunk c ends):
    // the first three of the five kh steps -- a load issued in the last step is still in flight at the barrier, and only the other resident workgroup
    // (none at all for the K-group instances) covers that wait.  Measured (tools/isteps_probe.sh, profiles/r04_isteps_probe.txt): 5 / 3 / 2 / 1 steps ->
    // 16.78 / 16.38 / 16.20 / 16.64 ms per-layer sum at batch 8, 8.94 / 8.60 / 8.79 / 9.12 at batch 4, dominant kernel 0.677 / 0.680 / 0.677 / 0.662 of the peak.
    // (Also tried there: s_setprio 1 around every 10-MFMA cluster -- 3 % slower -- and for the whole K loop -- neutral; tools/setprio_probe.sh,
    //  profiles/r04_setprio_probe.txt.  The two resident workgroups are whole tiles out of phase, not role-split waves: there is nothing to arbitrate.)
    constexpr int ISTEPS = (KS == 2 || WM == 8) ? 3 : NSTEP;
    static_assert(RA >= 1 && RA * NC * TT == S::N_BLK, "tile shape");
    static_assert(WROW % 4 == 0 && GPR % 4 == 0 && NP > (PPW - 1) * NW, "piece bookkeeping");

    const aid_conv2d_params& p = a.p;
    float* const sbuf0 = smem;
    float* const sbuf1 = smem + BUFSZ;
    float* const sbuf2 = smem + (NBUF == 3 ? 2 * BUFSZ : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = KS == 2 ? wave / NWQ : 0;             // K group of this wave
    const int wq = KS == 2 ? wave % NWQ : wave;
    const int wm = wq / WGN, wn = wq % WGN;
    const int half = lane >> 5;

    // XCD-aware logical tile (see conv53_wino4v_kernel); order: Cout tile fastest, then t tile, quad, residue-class group, sample
    const int Lt = (bid & 7) * ge.per_xcd + (bid >> 3);
    if (Lt >= ge.ntiles) return;
    const int nch = a.nchunks;                           // chunks this workgroup multiplies
    int rest = Lt;
    const int sp = SPK ? (rest & 1) : 0;                 // split-K: the two halves of a tile are neighbours in the same XCD's share
    if (SPK) rest >>= 1;
    const int tile_id = rest;
    const int by = rest % ge.ny; rest /= ge.ny;
    const int tile_t = rest % ge.ttiles; rest /= ge.ttiles;
    const int q = rest % ge.quads; rest /= ge.quads;
    const int rg = rest % ge.rgroups;
    const int b = rest / ge.rgroups;
    const int res = rg * NC;                             // first residue class of this tile
    const int t0 = tile_t * TT;
    const int m0 = ge.m_base + by * ge.m_stride;
    const int nrow = p.F / p.dilF;                       // rows of one residue class
    const int j0 = q * RA;                               // first sub-lattice row of this tile

    // ---- DMA piece descriptors ---------------------------------------------------------------------------------------------
    // A piece = 1 KiB of LDS filled by one wave instruction (16 bytes per lane).  Per piece and lane: ONE 32-bit byte offset from a wave-uniform
    // base (the activations or the weight pack, advanced by a scalar stride per chunk), and one bit of `pvalid`.  Lanes whose 16 bytes do not exist
    // (padding rows of the tile, groups past the end of a row, the tail of the last weight piece) are never loaded: their LDS slots are zeroed once
    // per tile in every buffer and the direct-to-LDS loads run with those lanes masked off; a piece none of whose lanes exists in this wave is not
    // issued at all (`pany`, wave-uniform, so that the per-chunk instruction count `mine` the vmcnt waits rely on stays exact).  (Round 3 kept a
    // 64-bit pointer and a per-lane stride per piece and read a zero page for the missing lanes: 27 more registers, which the 160-accumulator
    // F(8,3) instances do not have.)
    unsigned poff[PPW];
    unsigned pvalid = 0, pany = 0;
    int plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        poff[i] = 0; plds[i] = pc * 256;
        bool ok = false;
        if (pc < NXP) {
            const int eg = pc * 256 + 4 * lane;
            const int ci = eg / XCI, e = eg % XCI;
            const int gl = e % GPR, xi = (e / (IL * GPR)) % NXI;
            const int slot = (e / (IL * NXI * GPR)) * IL + (e / GPR) % IL;     // [slot / IL][xi][slot % IL][GPR]
            const int cls = slot / CSLOT;
            const int jr = j0 + (slot % CSLOT) - (KH / 2);   // sub-lattice row index of this slot
            if (ci < KCS && (slot % CSLOT) < CSLOT_USED && jr >= 0 && jr < nrow && t0 + WM * gl < p.T) {
                const int fi = res + cls + jr * p.dilF;
                poff[i] = (unsigned)(4 * ((int64_t)ci * p.x.sC + (int64_t)fi * p.x.sF + (int64_t)xi * (p.T / WM) + (t0 / WM) + gl));   // (within sample b: < 2^32, see conv53_wino_v)
                ok = true;
            }
        } else if (pc < NP) {
            const int e = (pc - NXP) * 256 + 4 * lane;
            const int row = e / WROW, col = e % WROW;
            const int tap = row / KCS, ci = row % KCS;
            plds[i] = XSZ + (pc - NXP) * 256;
            if (e < WSZ_RAW) {
                poff[i] = (unsigned)(4 * (((int64_t)tap * p.Cin_pad + ci) * p.Cout_pad + m0 + col));
                ok = true;
            }
        }
        if (ok) pvalid |= 1u << i;
        if (__ballot(ok) != 0ull) pany |= 1u << i;
    }
    const int cbase = (SPK && sp) ? a.nchunks : 0;        // first chunk of this workgroup's share of the K axis
    const int64_t xstep = (int64_t)KCS * p.x.sC * 4, wstep = (int64_t)KCS * p.Cout_pad * 4;   // bytes per chunk (wave-uniform)
    const char* const xbase = reinterpret_cast<const char*>(p.x.p + (int64_t)b * p.x.sB) + cbase * xstep;
    const char* const wbase = reinterpret_cast<const char*>(p.wp_wino) + cbase * wstep;
    // ---- operand addresses: lane's group g of this wave's 32 -> output row j = g / GPR, group tau = g % GPR -------------------------
    const int g = wn * 32 + (lane & 31);
    const int rl = g / GPR, tau = g % GPR;               // row of the tile: class rl / RA, sub-lattice row rl % RA
    const int cl = rl / RA, jl = rl % RA;
    const int sl0 = cl * CSLOT + jl;                     // slot of kh = 0; slot sl0 + kh: float offset of plane xi below
    auto xoff = [&](int slot, int xi) { return ((slot / IL) * NXI + xi) * (IL * GPR) + (slot % IL) * GPR; };
    int vBk[KH];                                         // per kh: offset of plane xi = 0 (planes are IL * GPR floats apart)
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) vBk[kh] = (ks * KC + half) * XCI + xoff(sl0 + kh, 0) + tau;
    const int vA = XSZ + (ks * KC + half) * WROW + wm * 32 + (lane & 31);

    f32x16 acc[NXI];
#pragma unroll
    for (int x = 0; x < NXI; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    const int mine = __builtin_popcount(pany);             // direct-to-LDS instructions this wave issues per chunk (exact)
    auto issue_piece = [&](auto ic, int ch, float* buf) {
        constexpr int i = decltype(ic)::value;
        if ((pany >> i) & 1u) {                             // (wave-uniform)
            const char* base = (wave + i * NW < NXP) ? xbase + ch * xstep : wbase + ch * wstep;     // (scalar)
            const unsigned la = AID_LDS_ADDR(buf + plds[i]);
            const unsigned off = poff[i];
            if ((pvalid >> i) & 1u) AID_DMA16_SBASE(off, base, la);
        }
    };
    // slots of the lanes that are never loaded: zero, in every buffer, before the first read
    aid_static_for<PPW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (wave + i * NW < NP && !((pvalid >> i) & 1u)) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sbuf0 + plds[i] + 4 * lane) = z;
            *reinterpret_cast<float4*>(sbuf1 + plds[i] + 4 * lane) = z;
            if (NBUF == 3) *reinterpret_cast<float4*>(sbuf2 + plds[i] + 4 * lane) = z;
        }
    });
    auto issue_all = [&](int ch, float* buf) { aid_static_for<PPW>([&](auto ic) { issue_piece(ic, ch, buf); }); };
    auto issue_step = [&](auto sc, int ch, float* buf) {
        aid_static_for<PPW>([&](auto ic) {
            if constexpr (decltype(sc)::value < ISTEPS && decltype(ic)::value % ISTEPS == decltype(sc)::value) issue_piece(ic, ch, buf);
        });
    };

    issue_all(0, sbuf0);
    if (NBUF == 3 && nch > 1) issue_all(1, sbuf1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = cur == 0 ? sbuf0 : (cur == 1 ? sbuf1 : sbuf2);
        // NBUF == 3: the loads of chunk ch+2 go to buffer (cur+2)%3 (read during the previous chunk); NBUF == 2: those of chunk ch+1 to the other buffer
        float* Nx = NBUF == 3 ? (cur == 0 ? sbuf2 : (cur == 1 ? sbuf0 : sbuf1)) : (cur == 0 ? sbuf1 : sbuf0);
        const int ahead = NBUF - 1;
        const bool more = (ch + ahead) < nch;
        float bv[2][NXI], av[2][NXI];
        auto load_step = [&](int kh, int buf) {
#pragma unroll
            for (int x = 0; x < NXI; ++x) bv[buf][x] = Bf[vBk[kh] + x * (IL * GPR)];
#pragma unroll
            for (int x = 0; x < NXI; ++x) av[buf][x] = Bf[vA + ((x * KH + kh) * KCS) * WROW];
        };
        load_step(0, 0);
        aid_static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (more) issue_step(sc, ch + ahead, Nx);
            if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
            constexpr int bq = s & 1;
#pragma unroll
            for (int x = 0; x < NXI; ++x)
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bq][x], bv[bq][x], acc[x], 0, 0, 0);
        });
        asm volatile("" ::: "memory");
        if (NBUF == 3 && more) {
            aid_wait_vmcnt_le<PPW>(mine);                   // the `mine` loads of chunk ch + 2 issued during this chunk may stay in flight
        } else {
            __builtin_amdgcn_s_waitcnt(AID_VMCNT(0));
        }
        __builtin_amdgcn_s_waitcnt(AID_LGKMCNT0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (NBUF == 3) {
        for (int ch = 0; ch < nch; ch += 3) {
            chunk(std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nch) chunk(std::integral_constant<int, 1>{}, ch + 1);
            if (ch + 2 < nch) chunk(std::integral_constant<int, 2>{}, ch + 2);
        }
    } else {
        for (int ch = 0; ch < nch; ch += 2) {
            chunk(std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nch) chunk(std::integral_constant<int, 1>{}, ch + 1);
        }
    }

    // ---- split-K: the workgroup that arrives first publishes its accumulators and leaves; the other adds them and runs the epilogue.
    // Arrival order is taken BEFORE anything is written, so the second workgroup only ever waits for one that is already running (no
    // dependence on dispatch order); a + b is commutative, so the sum does not depend on which half arrives first (deterministic).
    if constexpr (SPK) {
        // Exchange through agent-scope (sc1) loads and stores, which go past the per-XCD L2 to the coherent level, ordered by explicit
        // s_waitcnt + barrier: a release / acquire fence pair here writes back and invalidates the WHOLE L2 of the XCD once per tile
        // (measured: +45 us on a 150 us launch, the other tiles re-fetch their operands).
        unsigned* fl = a.flags + 2 * (int64_t)tile_id;
        unsigned long long* part = reinterpret_cast<unsigned long long*>(a.part + (int64_t)tile_id * (NXI * 16 * 64 * NW)) + tid;
        int* sh = reinterpret_cast<int*>(smem);
        if (tid == 0) sh[0] = (int)__hip_atomic_fetch_add(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int second = sh[0];
        if (!second) {
#pragma unroll
            for (int x = 0; x < NXI; ++x)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const unsigned long long v = (unsigned long long)__float_as_uint(acc[x][2 * r2]) | ((unsigned long long)__float_as_uint(acc[x][2 * r2 + 1]) << 32);
                    __hip_atomic_store(part + (x * 8 + r2) * (64 * NW), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this wave has reached the coherent level
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            // (bounded, ~0.2 s: the first workgroup is already past its K loop, this wait is microseconds; a bound keeps a logic error from hanging
            //  the device -- and if it is ever hit the tile is written as NaN, not as half a sum.  A timeout also sets the STICKY word at the end of
            //  the flag region: the late workgroup will still publish into flags that were reset under it, so every later split launch on this
            //  scratch writes NaN too instead of trusting a stale flag; the scratch has to be re-zeroed by the caller)
            unsigned* sticky = a.flags + (AID_W4R_SPLIT_FLAG_BYTES / 4 - 1);
            int it = 0;
            for (; it < (1 << 20) && __hip_atomic_load(fl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++it) __builtin_amdgcn_s_sleep(4);
            if (it >= (1 << 20)) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[1] = it < (1 << 20) && __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        }
        __syncthreads();
        const float poison = sh[1] ? 0.f : __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int x = 0; x < NXI; ++x)
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                const unsigned long long v = __hip_atomic_load(part + (x * 8 + r2) * (64 * NW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc[x][2 * r2] += __uint_as_float((unsigned)v) + poison; acc[x][2 * r2 + 1] += __uint_as_float((unsigned)(v >> 32)) + poison;
            }
        if (tid == 0) {                                   // leave the flags as they were found (the next launch on this stream reuses them)
            __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();                                  // sh[0] lives in the buffer the partial-sum reduction below reuses
    }

    // ---- K groups: the two groups hold partial sums of the SAME 64 x 512 tile.  Group 0 keeps accumulator rows 0..7 of every plane and takes the partner
    // lane's (same wave of the other group, same lane) rows 0..7, group 1 the same with rows 8..15 -- through LDS, five planes (80 KB) at a time -- so
    // each group then runs the epilogue of half the tile's output channels.  a + b is commutative: both orders give the same bits.
    if constexpr (KS == 2) {
        constexpr int GT = 64 * NWQ, HP = NXI / 2;
        static_assert(NXI % 2 == 0 && 2 * HP * 8 * GT <= S::LDS, "exchange region");
        float* const xo = smem + (ks == 0 ? 0 : HP * 8 * GT) + (tid & (GT - 1));        // written by this group
        const float* const xi_ = smem + (ks == 0 ? HP * 8 * GT : 0) + (tid & (GT - 1)); // written by the other one
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            if (ks == 0) {
#pragma unroll
                for (int x = 0; x < HP; ++x)
#pragma unroll
                    for (int r = 0; r < 8; ++r) xo[(x * 8 + r) * GT] = acc[h2 * HP + x][8 + r];
            } else {
#pragma unroll
                for (int x = 0; x < HP; ++x)
#pragma unroll
                    for (int r = 0; r < 8; ++r) xo[(x * 8 + r) * GT] = acc[h2 * HP + x][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int x = 0; x < HP; ++x)
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[h2 * HP + x][r] += xi_[(x * 8 + r) * GT];
            } else {                                       // (kept rows move down to 0..7: the epilogue below indexes registers statically)
#pragma unroll
                for (int x = 0; x < HP; ++x)
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[h2 * HP + x][r] = acc[h2 * HP + x][8 + r] + xi_[(x * 8 + r) * GT];
            }
            __syncthreads();
        }
    }

    // ---- epilogue (as conv53_wino4v_kernel; this lane's output row: sub-lattice row j0 + jl) -----------------------------------------
    // What the epilogue costs a launch (round 3, profiles/r03_w4r_epilogue_probe.txt, builds with an early return / without the dGELU
    // arithmetic): 2-11 % for the plain epilogue, 8-20 % for the dGELU one (3-7 % of it libm erff + expf) -- on ISOLATED launches.  Tried on
    // top of that: the four 4-row blocks software-pipelined (loads of block k+1 before block k is transformed and stored): dGELU launches
    // -7 ... +3 %, plain launches slower, forward-only evaluation -2 %; a 13-instruction dGELU (one exp shared by density and tail, A&S 7.1.26):
    // the isolated gap closes to 2 %, end to end 44.03 vs 44.05 evaluations/s -- under the sub-batch streams the epilogue arithmetic already runs
    // beneath other workgroups' MFMAs.  Neither is kept; nor are the experiment switches (they spill in the 168-register instances).
    float dsum[4] = {0.f, 0.f, 0.f, 0.f};                // <y, aux> (dot_ws) or sum y (stat_ws) per block of 4 rows
    float qsum[4] = {0.f, 0.f, 0.f, 0.f};                // sum y^2 (stat_ws)
    const int jr_o = j0 + jl;
    const int f_o = res + cl + jr_o * p.dilF;
    const int t_o = t0 + WM * tau;
    const bool ok_o = jr_o < nrow && t_o < p.T;
    const int mbase = m0 + wm * 32 + 4 * half;
    if constexpr (WM == 8) {
      if (ok_o) {
        // F(8,3): eight outputs per lane and accumulator row -- two rows per batch (the same bytes in flight as four rows of F(4,3))
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f_o * p.y.sF + t_o;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f_o * p.res.sF + t_o) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f_o * p.aux.sF + t_o) : 0;
#pragma unroll
        for (int r0 = 0; r0 < 16 / KS; r0 += 2) {          // (K groups: rows 0..7 of group k are accumulator rows 8 k .. 8 k + 7 of the tile)
            float rv[2][8], ur[2][8];
            float sv[2], as[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int r = r0 + qq;
                const int m = mbase + (r & 3) + 8 * ((r >> 2) + (KS == 2 ? 2 * ks : 0));
                const bool ok = m < p.Cout;
#pragma unroll
                for (int h4 = 0; h4 < 2; ++h4) {
                    const float4 v = (ok && p.res.p) ? *reinterpret_cast<const float4*>(p.res.p + rbase + (int64_t)m * p.res.sC + 4 * h4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rv[qq][4 * h4] = v.x; rv[qq][4 * h4 + 1] = v.y; rv[qq][4 * h4 + 2] = v.z; rv[qq][4 * h4 + 3] = v.w;
                }
                sv[qq] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                as[qq] = 0.f;
                if (ok && p.epi == 1) as[qq] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
#pragma unroll
                for (int h4 = 0; h4 < 2; ++h4) {
                    const float4 v = (ok && p.epi == 1) ? *reinterpret_cast<const float4*>(p.aux.p + abase + (int64_t)m * p.aux.sC + 4 * h4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    ur[qq][4 * h4] = v.x; ur[qq][4 * h4 + 1] = v.y; ur[qq][4 * h4 + 2] = v.z; ur[qq][4 * h4 + 3] = v.w;
                }
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int r = r0 + qq;
                const int m = mbase + (r & 3) + 8 * ((r >> 2) + (KS == 2 ? 2 * ks : 0));
                if (m >= p.Cout) continue;
                float Mv[NXI], y[8];
#pragma unroll
                for (int x = 0; x < NXI; ++x) Mv[x] = acc[x][r];
                aid_wino8_output(Mv, y);
                float ds = 0.f, ss = 0.f, qs = 0.f;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float v = y[o] * sv[qq];
                    if (p.epi == 1) v *= aid_dgelu(ur[qq][o] * as[qq]);
                    v += p.res_scale * rv[qq][o];
                    v *= p.alpha;
                    y[o] = v;
                    ds += v * ur[qq][o]; ss += v; qs += v * v;
                }
                float* yo = p.y.p + ybase + (int64_t)m * p.y.sC;
                *reinterpret_cast<float4*>(yo) = make_float4(y[0], y[1], y[2], y[3]);
                *reinterpret_cast<float4*>(yo + 4) = make_float4(y[4], y[5], y[6], y[7]);
                if (p.dot_ws) dsum[r >> 2] += ds;
                if (p.stat_ws) { dsum[r >> 2] += ss; qsum[r >> 2] += qs; }
            }
        }
      }
    } else
    if (ok_o) {
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f_o * p.y.sF + t_o;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f_o * p.res.sF + t_o) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f_o * p.aux.sF + t_o) : 0;
#pragma unroll
        for (int r0 = 0; r0 < 16 / KS; r0 += 4) {
            float4 rv[4], ur[4];
            float sv[4], as[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int r = r0 + qq;
                const int m = mbase + (r & 3) + 8 * ((r >> 2) + (KS == 2 ? 2 * ks : 0));
                const bool ok = m < p.Cout;
                rv[qq] = (ok && p.res.p) ? *reinterpret_cast<const float4*>(p.res.p + rbase + (int64_t)m * p.res.sC) : make_float4(0.f, 0.f, 0.f, 0.f);
                sv[qq] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                if (ok && p.epi == 1) {
                    as[qq] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                    ur[qq] = *reinterpret_cast<const float4*>(p.aux.p + abase + (int64_t)m * p.aux.sC);
                } else { as[qq] = 0.f; ur[qq] = make_float4(0.f, 0.f, 0.f, 0.f); }
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int r = r0 + qq;
                const int m = mbase + (r & 3) + 8 * ((r >> 2) + (KS == 2 ? 2 * ks : 0));
                if (m >= p.Cout) continue;
                const float M0 = acc[0][r], M1 = acc[1][r], M2 = acc[2][r], M3 = acc[3][r], M4 = acc[4][r], M5 = acc[5][r];
                const float a12 = M1 + M2, s12 = M1 - M2, a34 = M3 + M4, s34 = M3 - M4;
                float y0 = (M0 + a12 + a34) * sv[qq];
                float y1 = (s12 + 2.f * s34) * sv[qq];
                float y2 = (a12 + 4.f * a34) * sv[qq];
                float y3 = (s12 + 8.f * s34 + M5) * sv[qq];
                if (p.epi == 1) { y0 *= aid_dgelu(ur[qq].x * as[qq]); y1 *= aid_dgelu(ur[qq].y * as[qq]); y2 *= aid_dgelu(ur[qq].z * as[qq]); y3 *= aid_dgelu(ur[qq].w * as[qq]); }
                y0 += p.res_scale * rv[qq].x; y1 += p.res_scale * rv[qq].y; y2 += p.res_scale * rv[qq].z; y3 += p.res_scale * rv[qq].w;
                y0 *= p.alpha; y1 *= p.alpha; y2 *= p.alpha; y3 *= p.alpha;
                *reinterpret_cast<float4*>(p.y.p + ybase + (int64_t)m * p.y.sC) = make_float4(y0, y1, y2, y3);
                if (p.dot_ws) dsum[r0 >> 2] += (y0 * ur[qq].x + y1 * ur[qq].y) + (y2 * ur[qq].z + y3 * ur[qq].w);
                if (p.stat_ws) { dsum[r0 >> 2] += (y0 + y1) + (y2 + y3); qsum[r0 >> 2] += (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3); }
            }
        }
    }
    if (p.dot_ws || p.stat_ws) {                             // per (sample, channel group): one partial per tile
        const bool st = p.stat_ws != nullptr;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float v = dsum[qq], w = qsum[qq];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) { v += __shfl_xor(v, off, 32); if (st) w += __shfl_xor(w, off, 32); }
            dsum[qq] = v; qsum[qq] = w;
        }
        float* red = sbuf0;
        if ((lane & 31) == 0) {
            if constexpr (KS == 2) {                         // group k holds row blocks 2 k, 2 k + 1 (in its slots 0, 1)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) { red[(wq * 2 + half) * 4 + 2 * ks + qq] = dsum[qq]; red[NWQ * 8 + (wq * 2 + half) * 4 + 2 * ks + qq] = qsum[qq]; }
            } else {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { red[(wave * 2 + half) * 4 + qq] = dsum[qq]; red[NW * 8 + (wave * 2 + half) * 4 + qq] = qsum[qq]; }
            }
        }
        __syncthreads();
        const int cpg = p.Cout >> 3;
        const int grp = a.dot_all ? tid : m0 / cpg + tid;    // dot_all: zeros for the groups this Cout tile does not touch
        if (tid < (a.dot_all ? 8 : M_BLK / cpg) && grp < 8) {
            float sacc = 0.f, qacc = 0.f;
            for (int w = 0; w < NWQ; ++w)
                for (int h = 0; h < 2; ++h)
                    for (int qq = 0; qq < 4; ++qq) {
                        const int mrow = m0 + (w / WGN) * 32 + 4 * h + 8 * qq;
                        if (mrow < p.Cout && mrow / cpg == grp) { sacc += red[(w * 2 + h) * 4 + qq]; qacc += red[NWQ * 8 + (w * 2 + h) * 4 + qq]; }
                    }
            const int ptile = ge.dot_base + (rg * ge.quads + q) * ge.ttiles + tile_t;
            if (st) {
                double* o = p.stat_ws + (((int64_t)b * 8 + grp) * p.stat_n + ptile) * 2;
                if (p.fin_mode) { aid_st_agent(o, (double)sacc); aid_st_agent(o + 1, (double)qacc); }
                else { o[0] = (double)sacc; o[1] = (double)qacc; }
            } else {
                double* o = p.dot_ws + ((int64_t)b * 8 + grp) * p.dot_n + ptile;
                if (p.fin_mode) aid_st_agent(o, (double)sacc);
                else *o = (double)sacc;
            }
        }
        if (p.fin_mode) {                                    // (wave-uniform) the last tile of sample b folds the sample's partials: aid_wino_fin
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's partials have reached the coherent level
            __syncthreads();
            int* shi = reinterpret_cast<int*>(sbuf0) + 2 * NW * 8;      // (behind `red`)
            if (tid == 0) shi[0] = (__hip_atomic_fetch_add(p.fin_count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.fin_total - 1)) ? 1 : 0;
            __syncthreads();
            if (shi[0]) {
                __syncthreads();                               // (everybody has read shi[0]: the scratch below overlaps it)
                const double n = (double)(p.Cout >> 3) * (double)p.F * (double)p.T;
                aid_wino_fin(p.fin_mode, st ? p.stat_ws : p.dot_ws, st ? p.stat_n : p.dot_n, b, p.Cout, n, p.fin_gamma, p.fin_mod, p.fin_mod_ld, p.fin_eps,
                             p.fin_scale, p.fin_stats, reinterpret_cast<double*>(sbuf0), tid, 64 * NW);
                if (tid == 0) __hip_atomic_store(p.fin_count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // as found, for the next launch
            }
        }
    }
}

// NC1 > 0: PAIR instance for 96-channel layers -- the 64 x 256 tiles of channels [0, 64) (family 0: <TT, NC, WGM = 2>) and the 32 x 512
// remainder tiles of channels [64, 96) (family 1: <TT, NC1, WGM = 1>) in ONE grid, family 0 first: the remainder tiles start while the
// last 64-wide tiles drain instead of after a launch boundary (two launches: each has its own tail, and at batch 1 neither fills the chip).
// Measured against two launches (profiles/r03_w4r_split_probe.txt, r03_w4r_ab.txt): batch 1 [96, 256, 256] 140 -> 100 us,
// [96, 192, 512] 170 -> 145 us; batch 3 380 -> 340 us; batch 8 918 -> 905 / 627 -> 604 us; end to end +4.7 % / +1.5 % / +1.2 % / +0.6 % at batch 1 / 2 / 3 / 8.  (The two-launch path was removed after that A/B.)
// SPK: split-K instance (ConvWinoRDev::splits == 2).
template <int TT, int NC, int WGM, int NB = 3, int WPC = 2, int NWV = 4, int NC1 = 0, bool SPK = false>
__global__ __launch_bounds__(64 * NWV, (WPC * NWV + 3) / 4) void conv53_wino4r_kernel(const ConvWinoRDev a) {
    using S0 = W4rShape<TT, NC, WGM, NB, WPC, NWV>;
    if constexpr (NC1 == 0) {
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS];
        wino4r_tile_body<TT, NC, WGM, NB, WPC, NWV, SPK>(a, a.g[0], smem, (int)blockIdx.x);
    } else {
        using S1 = W4rShape<TT, NC1, 1, NB, WPC, NWV>;
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS > S1::LDS ? S0::LDS : S1::LDS];
        const int n0 = 8 * a.g[0].per_xcd;
        if ((int)blockIdx.x < n0) wino4r_tile_body<TT, NC, WGM, NB, WPC, NWV, false>(a, a.g[0], smem, (int)blockIdx.x);
        else wino4r_tile_body<TT, NC1, 1, NB, WPC, NWV, false>(a, a.g[1], smem, (int)blockIdx.x - n0);
    }
}

// K-group instance of the F(4,3) kernel (see conv53_wino8r_ks_kernel below): one eight-wave workgroup per 64 x 256 tile, two K groups, for launches of at
// most one tile per CU -- the deepest level of the shipped networks at batch 1 (224 tiles).  Replaces the split-K instances there: the same two waves
// per SIMD, the partial sums through LDS instead of 98 KB per tile through memory (141-159 -> see DESIGN.md 3.1e).
template <int TT, int NC>
__global__ __launch_bounds__(512, 2) void conv53_wino4r_ks_kernel(const ConvWinoRDev a) {
    using S0 = W4rShape<TT, NC, 2, 3, 1, 8, 4, 2>;
    __shared__ __attribute__((aligned(16))) float smem[S0::LDS];
    wino4r_tile_body<TT, NC, 2, 3, 1, 8, false, 4, 2>(a, a.g[0], smem, (int)blockIdx.x);
}

// ---- F(8,3) on the same row-shared body: 10 products per 8 outputs (0.833x the MFMAs of F(4,3), 0.417x of the direct form), Winograd-domain
// input [B, C, F, 10, T/8] (1.25x the activation instead of 1.5x).  A tile is 64 output channels x 512 positions (64 groups: 160 accumulator
// registers per lane), 33-36 KB per LDS buffer (50 transformed taps x 2 channels x 64 = 25 KB of weights), two buffers, two workgroups per CU.
// fp32 error 3e-6 rel-L2 per layer at Cin = 128 (F(4,3): 1.5e-6, direct: 0.8e-6; tools/wino_fm3_error.py, profiles/r04_wino_fm3_error.txt).
// NC1 > 0: pair instance for the 96-channel layers (32 x 1024 remainder tiles), as above.
template <int TT, int NC, int WGM, int NC1 = 0>
__global__ __launch_bounds__(256, 2) void conv53_wino8r_kernel(const ConvWinoRDev a) {
    using S0 = W4rShape<TT, NC, WGM, 2, 2, 4, 8>;
    if constexpr (NC1 == 0) {
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS];
        wino4r_tile_body<TT, NC, WGM, 2, 2, 4, false, 8>(a, a.g[0], smem, (int)blockIdx.x);
    } else {
        using S1 = W4rShape<TT, NC1, 1, 2, 2, 4, 8>;
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS > S1::LDS ? S0::LDS : S1::LDS];
        const int n0 = 8 * a.g[0].per_xcd;
        if ((int)blockIdx.x < n0) wino4r_tile_body<TT, NC, WGM, 2, 2, 4, false, 8>(a, a.g[0], smem, (int)blockIdx.x);
        else wino4r_tile_body<TT, NC1, 1, 2, 2, 4, false, 8>(a, a.g[1], smem, (int)blockIdx.x - n0);
    }
}

// K-group instance of the F(8,3) kernel: ONE eight-wave workgroup per CU, the same 64 x 512 (32 x 1024) tile, waves 0-3 on the even and waves 4-7 on the
// odd pairs of input channels (four channels per staged chunk, 66-72 KB per buffer, two buffers).  The F(8,3) tile has 160 accumulators per lane, so a
// 64 x 512 tile is ONE wave per SIMD; a launch with no more tiles than CUs (batch 1: 256 on the upper levels, 192 / 224 on the deepest) then runs with the
// matrix pipe waiting on every LDS read and barrier.  Two waves per SIMD on the SAME tile need no second tile; the exchange at the end is 160 KB of
// LDS traffic per tile (about 1 us).
template <int TT, int NC, int WGM, int NC1 = 0>
__global__ __launch_bounds__(512, 2) void conv53_wino8r_ks_kernel(const ConvWinoRDev a) {
    using S0 = W4rShape<TT, NC, WGM, 2, 1, 8, 8, 2>;
    if constexpr (NC1 == 0) {
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS];
        wino4r_tile_body<TT, NC, WGM, 2, 1, 8, false, 8, 2>(a, a.g[0], smem, (int)blockIdx.x);
    } else {
        using S1 = W4rShape<TT, NC1, 1, 2, 1, 8, 8, 2>;
        __shared__ __attribute__((aligned(16))) float smem[S0::LDS > S1::LDS ? S0::LDS : S1::LDS];
        const int n0 = 8 * a.g[0].per_xcd;
        if ((int)blockIdx.x < n0) wino4r_tile_body<TT, NC, WGM, 2, 1, 8, false, 8, 2>(a, a.g[0], smem, (int)blockIdx.x);
        else wino4r_tile_body<TT, NC1, 1, 2, 1, 8, false, 8, 2>(a, a.g[1], smem, (int)blockIdx.x - n0);
    }
}

template <int MT, int NTT, int WGM, int WGN, int RMAX, int KC, int MINW = 1>
static int launch_wino4(const aid_conv2d_params* p, hipStream_t st) {
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 128 * NTT * WGN;
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page_w)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    ConvWinoDev a;
    a.p = *p;
    a.zero = zero;
    int TT = aid_pow2ceil(p->T);
    if (TT > N_BLK) TT = N_BLK;
    a.tt_log2 = aid_ilog2(TT);
    const int ROWS = N_BLK / TT;
    if (ROWS > RMAX) return 1000;
    a.rows_log2 = aid_ilog2(ROWS);
    a.tiles_t = aid_cdiv(p->T, TT);
    a.nrows = p->B * p->F;
    a.nchunks = p->Cin / KC;
    const int rgroups = aid_cdiv(a.nrows, ROWS);
    dim3 grid((unsigned)(rgroups * a.tiles_t), (unsigned)(p->Cout_pad / M_BLK));
    auto kern = conv53_wino4_kernel<MT, NTT, WGM, WGN, RMAX, KC, MINW>;      // LDS is static (see the kernel)
    a.nx = (int)grid.x; a.ny = (int)grid.y;
    a.per_xcd = (a.nx * a.ny + 7) / 8;                  // XCD-aware 1-D grid (see the kernel)
    grid = dim3((unsigned)(8 * a.per_xcd), 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel(M_BLK == 96 ? "conv53_wino4_kernel(96)" : ((M_BLK == 64 && N_BLK == 256) ? "conv53_wino4_kernel(64x256)" : "conv53_wino4_kernel"));
    return AID_OK;
}

template <int MT, int NTT, int WGM, int WGN, int RMAX, int KC, int MINW = 1>
static int launch_wino4v(const aid_conv2d_params* p, hipStream_t st) {
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 128 * NTT * WGN;
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page_w)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    ConvWinoDev a;
    a.p = *p;
    a.zero = zero;
    int TT = aid_pow2ceil(p->T);
    if (TT > N_BLK) TT = N_BLK;
    a.tt_log2 = aid_ilog2(TT);
    const int ROWS = N_BLK / TT;
    if (ROWS > RMAX) return 1000;
    a.rows_log2 = aid_ilog2(ROWS);
    a.tiles_t = aid_cdiv(p->T, TT);
    a.nrows = p->B * p->F;
    a.nchunks = p->Cin / KC;
    const int rgroups = aid_cdiv(a.nrows, ROWS);
    dim3 grid((unsigned)(rgroups * a.tiles_t), (unsigned)(p->Cout_pad / M_BLK));
    auto kern = conv53_wino4v_kernel<MT, NTT, WGM, WGN, RMAX, KC, MINW>;      // LDS is static (see the kernel)
    a.nx = (int)grid.x; a.ny = (int)grid.y;
    a.per_xcd = (a.nx * a.ny + 7) / 8;                  // XCD-aware 1-D grid (see the kernel)
    grid = dim3((unsigned)(8 * a.per_xcd), 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel((M_BLK == 64 && N_BLK == 256) ? "conv53_wino4v_kernel(64x256)" : (M_BLK == 96 ? "conv53_wino4v_kernel(96)" : "conv53_wino4v_kernel"));
    return AID_OK;
}

// row-shared kernel: eligibility + launch.  Returns 1000 when the shape is better served by the 64|96 x 512 kernel.
// Picks the fewest residue classes per tile (least staging) whose rows-per-class RA wastes at most 10 % of a tile on padding rows.
// npos: positions per tile (256 for the 64-channel tile, 512 for the 32-channel remainder tile of 96-channel layers).
static int wino4r_tile(const aid_conv2d_params* p, int npos, int max_nc, int* TTo, int* NCo, int* quads, int* ttiles) {
    if (p->dilF < 1 || (p->F % p->dilF)) return 0;
    const int TT = p->T >= 64 ? 64 : (p->T >= 32 ? 32 : 16);      // (T = 16: the deepest level of the 8-octave 44.1 kHz network at L = 184184)
    if (p->T % TT) return 0;
    const int nrow = p->F / p->dilF;
    for (int NC = 1; NC * TT <= 256 && NC <= max_nc; NC *= 2) {
        if (p->dilF % NC) break;
        const int RA = npos / (TT * NC);
        const int qd = (nrow + RA - 1) / RA;
        if ((int64_t)qd * RA * 10 > (int64_t)nrow * 11) continue;     // more than 10 % of the rows of a tile would be padding
        *TTo = TT; *NCo = NC; *quads = qd; *ttiles = p->T / TT;
        return 1;
    }
    return 0;
}

struct Wino4rPlan { int TT, NC, quads, ttiles; };
// plan[0]: the 64-channel tiles; plan[1] (96-channel layers only): the 32-channel remainder tiles.  Returns the number of launches (0: not eligible).
static int wino4r_geometry(const aid_conv2d_params* p, Wino4rPlan plan[2]) {
    if (p->Cout_pad % 64 == 0) return wino4r_tile(p, 256, 8, &plan[0].TT, &plan[0].NC, &plan[0].quads, &plan[0].ttiles) ? 1 : 0;
    if (p->Cout_pad % 96) return 0;
    if (!wino4r_tile(p, 256, 8, &plan[0].TT, &plan[0].NC, &plan[0].quads, &plan[0].ttiles)) return 0;
    if (!wino4r_tile(p, 512, 2, &plan[1].TT, &plan[1].NC, &plan[1].quads, &plan[1].ttiles) || plan[1].TT != 64) return 0;
    return 2;
}

static bool wino_v_shape_ok(int Cin, int Cout, int T);

// (environment overrides exist only in -DAID_EXPERIMENT builds: the product library's results never depend on the environment)
static int w4r_env(const char* name, int dflt) {
#ifdef AID_EXPERIMENT
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Split-K of the row-shared kernel (batch 1 with the scratch `ws` given -- which the network's launch plans do only for a WHOLE batch of one, so
// that the result of a sample never depends on the sub-batch split it is evaluated in): a launch with fewer tiles than CUs -- 192 / 224 on the deepest levels of the 22.05 kHz network -- gives
// every tile to TWO workgroups, one per half of the input channels: one wave per SIMD keeps the matrix pipe 76 % busy, two 91 %.
// The tile count per CU does not change (the launch stays bound by 1 tile's MFMAs per CU), so the gain is that utilisation only:
// 155 -> 147 us (d = 1) ... 169 -> 143 us (d = 64) per launch on [256, 448, 32]; launches with 256 < tiles <= 336 measured neutral to -4 %
// (profiles/r03_w4r_split_probe.txt); end to end at batch 1: 27.75 -> 28.37 evaluations/s with the threshold at 230, 28.22 at 336
// (profiles/r03_w4r_ab.txt).  AID_W4R_SPLIT overrides the threshold (0 = never split).
// K-group instances of the F(4,3) kernel (conv53_wino4r_ks_kernel): at most one tile per CU, the T <= 32 tile shapes (the deep levels).
// AID_W4R_KS (experiment builds): 0 never (split-K as before), 1 wherever instantiated.
static bool wino4r_ks_wanted(const aid_conv2d_params* p, int nl, const Wino4rPlan* plan, int64_t ntiles) {
    static const int force = w4r_env("AID_W4R_KS", -1);
    if (nl != 1 || plan[0].TT > 32 || (p->Cin % 4)) return false;
    if (force == 0 || force == 1) return force != 0;
    return ntiles <= 256;
}
static int wino4r_splits(const aid_conv2d_params* p, int nl, int64_t ntiles) {
    static const int thr = w4r_env("AID_W4R_SPLIT", 230);
    if (p->B != 1 || nl != 1 || ntiles > thr || ntiles > AID_W4R_SPLIT_MAX_TILES || (p->Cin % 4)) return 1;
    return 2;
}

// fin_mode (aid_kernels.h): argument check shared by the two row-shared launchers
static int wino_fin_check(const aid_conv2d_params* p) {
    if (!p->fin_mode) return AID_OK;
    AID_REQUIRE(p->fin_mode == 1 || p->fin_mode == 2, "aid_conv2d: fin_mode is 0, 1 or 2");
    AID_REQUIRE(p->fin_count && p->fin_scale && (p->Cout % 8) == 0, "aid_conv2d: fin_mode needs fin_count, fin_scale and Cout % 8 == 0");
    if (p->fin_mode == 1) AID_REQUIRE(p->stat_ws && p->fin_gamma, "aid_conv2d: fin_mode = 1 folds the stat_ws partials and needs fin_gamma");
    else AID_REQUIRE(p->dot_ws && p->fin_stats, "aid_conv2d: fin_mode = 2 folds the dot_ws partials and needs the forward statistics in fin_stats");
    return AID_OK;
}

static int launch_wino4r(const aid_conv2d_params* p, hipStream_t st) {
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page_w)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    Wino4rPlan plan[2];
    const int nl = wino4r_geometry(p, plan);
    if (!nl) return 1000;
    const bool m96 = nl == 2;
    ConvWinoRDev a;
    a.p = *p;
    a.zero = zero;
    a.nchunks = p->Cin / 2;
    a.dot_all = m96 ? 1 : 0;
    a.splits = 1; a.part = nullptr; a.flags = nullptr;
    W4rGeo geo[2];
    int dot_base = 0;
    for (int l = 0; l < nl; ++l) {
        const Wino4rPlan& g = plan[l];
        W4rGeo& ge = geo[l];
        ge.quads = g.quads; ge.ttiles = g.ttiles;
        ge.rgroups = p->dilF / g.NC;
        ge.ny = m96 ? p->Cout_pad / 96 : p->Cout_pad / 64;
        ge.m_base = l == 0 ? 0 : 64;
        ge.m_stride = m96 ? 96 : 64;
        ge.dot_base = dot_base;
        dot_base += ge.rgroups * g.quads * g.ttiles;
        ge.ntiles = p->B * ge.rgroups * g.quads * g.ttiles * ge.ny;
        ge.per_xcd = (ge.ntiles + 7) / 8;
    }
    { const int rc = wino_fin_check(p); if (rc != AID_OK) return rc; }
    a.fin_total = (geo[0].ntiles + (m96 ? geo[1].ntiles : 0)) / p->B;      // tiles per sample (before a split doubles the workgroups)
    const bool kg = wino4r_ks_wanted(p, nl, plan, geo[0].ntiles);
    if (kg) {
        a.nchunks = p->Cin / 4;
    } else if (wino4r_splits(p, nl, geo[0].ntiles) == 2 && p->ws && p->ws_bytes >= aid_w4r_split_bytes(geo[0].ntiles)) {
        a.splits = 2;
        a.flags = reinterpret_cast<unsigned*>(p->ws);
        a.part = p->ws + AID_W4R_SPLIT_FLAG_BYTES / 4;
        a.nchunks = p->Cin / 4;
        geo[0].ntiles *= 2;
        geo[0].per_xcd = (geo[0].ntiles + 7) / 8;
    }
    const bool short_k = p->Cin <= 128 && a.splits == 1;     // three workgroups per CU (two LDS buffers) for the short-K layers (re-measured in round 3 with
                                                             // sub-batch streams: Cin <= 128 / 64 / never = 42.33 / 42.08 / 42.17 evaluations/s)
#define AID_W4R(TTv, NCv, WGMv, NC1v) do { \
        if constexpr (TTv == 64 && NCv <= 2) { if (short_k) { hipLaunchKernelGGL((conv53_wino4r_kernel<TTv, NCv, WGMv, 2, 3, 4, NC1v>), grid, dim3(256), 0, st, a); break; } } \
        if constexpr (NC1v == 0 && WGMv == 2) { if (a.splits == 2) { hipLaunchKernelGGL((conv53_wino4r_kernel<TTv, NCv, WGMv, 3, 2, 4, 0, true>), grid, dim3(256), 0, st, a); break; } } \
        hipLaunchKernelGGL((conv53_wino4r_kernel<TTv, NCv, WGMv, 3, 2, 4, NC1v>), grid, dim3(256), 0, st, a); } while (0)
    if (m96) {
        a.g[0] = geo[0]; a.g[1] = geo[1];
        const dim3 grid((unsigned)(8 * (geo[0].per_xcd + geo[1].per_xcd)));
        switch (plan[0].NC * 16 + plan[1].NC) {               // (both families: TT = 64, wino4r_geometry)
            case 1 * 16 + 1: AID_W4R(64, 1, 2, 1); break;
            case 1 * 16 + 2: AID_W4R(64, 1, 2, 2); break;
            case 2 * 16 + 1: AID_W4R(64, 2, 2, 1); break;
            case 2 * 16 + 2: AID_W4R(64, 2, 2, 2); break;
            case 4 * 16 + 1: AID_W4R(64, 4, 2, 1); break;
            case 4 * 16 + 2: AID_W4R(64, 4, 2, 2); break;
            default: aid_set_error("aid_conv2d: row-shared pair tile shape not instantiated"); return AID_E_BADARG;
        }
        AID_CHECK_LAUNCH();
    } else {
        const Wino4rPlan& g = plan[0];
        a.g[0] = geo[0]; a.g[1] = geo[0];
        const dim3 grid((unsigned)(8 * geo[0].per_xcd));
#define AID_W4R_KS(TTv, NCv) hipLaunchKernelGGL((conv53_wino4r_ks_kernel<TTv, NCv>), grid, dim3(512), 0, st, a)
        if (kg) switch (g.TT * 16 + g.NC) {
            case 32 * 16 + 1: AID_W4R_KS(32, 1); break;
            case 32 * 16 + 2: AID_W4R_KS(32, 2); break;
            case 32 * 16 + 4: AID_W4R_KS(32, 4); break;
            case 32 * 16 + 8: AID_W4R_KS(32, 8); break;
            case 16 * 16 + 1: AID_W4R_KS(16, 1); break;
            case 16 * 16 + 2: AID_W4R_KS(16, 2); break;
            case 16 * 16 + 4: AID_W4R_KS(16, 4); break;
            case 16 * 16 + 8: AID_W4R_KS(16, 8); break;
            default: aid_set_error("aid_conv2d: row-shared K-group tile shape not instantiated"); return AID_E_BADARG;
        } else
        switch (g.TT * 16 + g.NC) {
            case 64 * 16 + 1: AID_W4R(64, 1, 2, 0); break;
            case 64 * 16 + 2: AID_W4R(64, 2, 2, 0); break;
            case 64 * 16 + 4: AID_W4R(64, 4, 2, 0); break;
            case 32 * 16 + 1: AID_W4R(32, 1, 2, 0); break;
            case 32 * 16 + 2: AID_W4R(32, 2, 2, 0); break;
            case 32 * 16 + 4: AID_W4R(32, 4, 2, 0); break;
            case 32 * 16 + 8: AID_W4R(32, 8, 2, 0); break;
            case 16 * 16 + 1: AID_W4R(16, 1, 2, 0); break;
            case 16 * 16 + 2: AID_W4R(16, 2, 2, 0); break;
            case 16 * 16 + 4: AID_W4R(16, 4, 2, 0); break;
            case 16 * 16 + 8: AID_W4R(16, 8, 2, 0); break;
            default: aid_set_error("aid_conv2d: row-shared tile shape not instantiated"); return AID_E_BADARG;
        }
        AID_CHECK_LAUNCH();
    }
#undef AID_W4R
#undef AID_W4R_KS
    if (kg) aid_note_kernel("conv53_wino4r_ks_kernel");
    else aid_note_kernel(m96 ? "conv53_wino4r_kernel(64+32)" : (a.splits == 2 ? "conv53_wino4r_kernel(split-K)" : (plan[0].NC == 1 ? "conv53_wino4r_kernel" : "conv53_wino4r_kernel(multi-class)")));
    return AID_OK;
}

// ---- F(8,3) row-shared tiles: geometry, eligibility, launch ---------------------------------------------------------------------------------------
// ng: groups per tile (64 for the 64-channel tile = 512 positions, 128 for the 32-channel remainder tile of 96-channel layers).  Instantiated:
// 64 groups -- TT = 64: NC 1, 2 (RA = 8, 4); TT = 32: NC 1, 2, 4 (RA = 16, 8, 4); 128 groups -- TT = 64: NC 1, 2, 4 (RA = 16, 8, 4).  More classes per tile would not
// leave room for two workgroups per CU (NC (RA + 4) staged rows x 10 planes).
static int wino8r_tile(const aid_conv2d_params* p, int ng, Wino4rPlan* o) {
    if (p->dilF < 1 || (p->F % p->dilF) || (p->T % 32)) return 0;
    const int TT = (p->T % 64 == 0) ? 64 : 32;
    if (ng == 128 && TT != 64) return 0;
    const int GPR = TT / 8;
    const int nrow = p->F / p->dilF;
    const int max_nc = (ng == 64 && TT == 64) ? 2 : 4;
    for (int NC = 1; NC <= max_nc; NC *= 2) {
        if (p->dilF % NC) break;
        const int RA = ng / (GPR * NC);
        const int qd = (nrow + RA - 1) / RA;
        if ((int64_t)qd * RA * 100 > (int64_t)nrow * 115) continue;     // (up to 15 % padding rows: 7, 14, 28 rows per class -> 8, 16, 32; the form choice weighs them)
        o->TT = TT; o->NC = NC; o->quads = qd; o->ttiles = p->T / TT;
        return 1;
    }
    return 0;
}
static int wino8r_geometry(const aid_conv2d_params* p, Wino4rPlan plan[2]) {
    if (p->Cout_pad % 64 == 0) return wino8r_tile(p, 64, &plan[0]) ? 1 : 0;
    if (p->Cout_pad % 96) return 0;
    if (!wino8r_tile(p, 64, &plan[0]) || plan[0].TT != 64) return 0;
    if (!wino8r_tile(p, 128, &plan[1])) return 0;
    return 2;
}

// Which Winograd form the row-shared kernels want for a 5x3 layer of this LAUNCH shape: 8 (F(8,3)), 4 (F(4,3)) or 0 (no Winograd-domain input).
// F(8,3) issues 10 MFMAs per 8 outputs against 12, but its tiles hold twice the positions, so a launch has half as many of them.  The two
// resident workgroups of a CU share one matrix pipe: the busiest CU works through ceil(tiles / CUs) tiles, and a CU that only ever holds one tile
// runs it at about 5/6 of the rate two would reach.  cost = ceil(tiles / 256) x [MFMAs per tile: 5 vs 3, padding rows included in `tiles`]
// (x 1.2 for a single tile per CU -- x 1.03 for F(8,3), whose K-group instances put eight waves on that tile); the cheaper form takes the launch, ties go to F(4,3) (its smaller tiles leave the shorter tail).  Against the
// per-layer A/B of both kernels at batch 1, 2, 3, 4 and 8 (profiles/r04_streamk_probe.txt) this picks the faster kernel on all but a handful of
// shapes, where it is within 5 %.  A function of the launch shape, B included: equal launches (the same sub-batch size) take equal kernels.
static int wino_form_choice(const aid_conv2d_params* p) {
    if (!wino_v_shape_ok(p->Cin, p->Cout, p->T)) return 0;
    Wino4rPlan p4[2], p8[2];
    const int n4 = wino4r_geometry(p, p4);
    const int n8 = (p->T % 32 == 0) ? wino8r_geometry(p, p8) : 0;
    if (!n8) return (n4 || p->T >= 32) ? 4 : 0;
    if (!n4) return 8;
    auto cost = [&](const Wino4rPlan* pl, int nl, int per_tile) {
        int64_t tiles = 0;
        for (int l = 0; l < nl; ++l)
            tiles += (int64_t)p->B * (p->dilF / pl[l].NC) * pl[l].quads * pl[l].ttiles * (nl == 2 ? p->Cout_pad / 96 : p->Cout_pad / 64);
        const int64_t per_cu = (tiles + 255) / 256;
        // (launches of one tile per CU take the K-group instances where there is one: two waves per SIMD on the one tile, 1.03 instead of 1.2)
        const bool kgroups = per_tile == 5 || (nl == 1 && pl[0].TT <= 32);     // (wino8r_ks_wanted / wino4r_ks_wanted)
        return per_cu * per_tile * (per_cu == 1 ? (kgroups ? 103 : 120) : 100);
    };
    return cost(p8, n8, 5) < cost(p4, n4, 3) ? 8 : 4;
}

// K-group instances (conv53_wino8r_ks_kernel): a launch none of whose CUs would hold two tiles.  Per-layer A/B against the plain F(8,3) tiles and
// F(4,3) (profiles/r04_ks_probe.txt): -12 ... -18 % on the <= 256-tile launches (batch 1: levels 0, 3, 4, 5; batch 2: level 6), within +-2 % of the
// plain tiles elsewhere (where they are not used: the hardware's dynamic dispatch of two independent workgroups per CU is the safer schedule).
// AID_W8R_KS (experiment builds): 0 never, 1 always, 2 also the launches of 2 r + 1 half-rounds (512 < tiles <= 768).
static bool wino8r_ks_wanted(int64_t ntiles) {
    static const int force = w4r_env("AID_W8R_KS", -1);
    if (force == 0 || force == 1) return force != 0;
    if (force == 2 && ntiles > 512 && ntiles <= 768) return true;
    return ntiles <= 256;
}


static int launch_wino8r(const aid_conv2d_params* p, hipStream_t st) {
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page_w)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    Wino4rPlan plan[2];
    const int nl = wino8r_geometry(p, plan);
    AID_REQUIRE(nl, "aid_conv2d: x_wino = 2 (F(8,3)) needs a shape aid_conv2d_wino8_supported() answers 1 for");
    const bool m96 = nl == 2;
    ConvWinoRDev a;
    a.p = *p;
    a.zero = zero;
    a.nchunks = p->Cin / 2;
    a.dot_all = m96 ? 1 : 0;
    a.splits = 1; a.part = nullptr; a.flags = nullptr;
    W4rGeo geo[2];
    int dot_base = 0;
    for (int l = 0; l < nl; ++l) {
        const Wino4rPlan& g = plan[l];
        W4rGeo& ge = geo[l];
        ge.quads = g.quads; ge.ttiles = g.ttiles;
        ge.rgroups = p->dilF / g.NC;
        ge.ny = m96 ? p->Cout_pad / 96 : p->Cout_pad / 64;
        ge.m_base = l == 0 ? 0 : 64;
        ge.m_stride = m96 ? 96 : 64;
        ge.dot_base = dot_base;
        dot_base += ge.rgroups * g.quads * g.ttiles;
        ge.ntiles = p->B * ge.rgroups * g.quads * g.ttiles * ge.ny;
        ge.per_xcd = (ge.ntiles + 7) / 8;
    }
    const int64_t ntl = (int64_t)geo[0].ntiles + (m96 ? geo[1].ntiles : 0);
    { const int rc = wino_fin_check(p); if (rc != AID_OK) return rc; }
    a.fin_total = (int)(ntl / p->B);                       // tiles per sample
    const bool kg = wino8r_ks_wanted(ntl);
    if (kg) a.nchunks = p->Cin / 4;
#define AID_W8R(TTv, NCv, WGMv, NC1v) do { \
        if (kg) hipLaunchKernelGGL((conv53_wino8r_ks_kernel<TTv, NCv, WGMv, NC1v>), grid, dim3(512), 0, st, a); \
        else hipLaunchKernelGGL((conv53_wino8r_kernel<TTv, NCv, WGMv, NC1v>), grid, dim3(256), 0, st, a); } while (0)
    if (m96) {
        a.g[0] = geo[0]; a.g[1] = geo[1];
        const dim3 grid((unsigned)(8 * (geo[0].per_xcd + geo[1].per_xcd)));
        switch (plan[0].NC * 16 + plan[1].NC) {               // (both families: TT = 64)
            case 1 * 16 + 1: AID_W8R(64, 1, 2, 1); break;
            case 1 * 16 + 2: AID_W8R(64, 1, 2, 2); break;
            case 1 * 16 + 4: AID_W8R(64, 1, 2, 4); break;
            case 2 * 16 + 1: AID_W8R(64, 2, 2, 1); break;
            case 2 * 16 + 2: AID_W8R(64, 2, 2, 2); break;
            case 2 * 16 + 4: AID_W8R(64, 2, 2, 4); break;
            default: aid_set_error("aid_conv2d: F(8,3) pair tile shape not instantiated"); return AID_E_BADARG;
        }
        AID_CHECK_LAUNCH();
    } else {
        const Wino4rPlan& g = plan[0];
        a.g[0] = geo[0]; a.g[1] = geo[0];
        const dim3 grid((unsigned)(8 * geo[0].per_xcd));
        switch (g.TT * 16 + g.NC) {
            case 64 * 16 + 1: AID_W8R(64, 1, 2, 0); break;
            case 64 * 16 + 2: AID_W8R(64, 2, 2, 0); break;
            case 32 * 16 + 1: AID_W8R(32, 1, 2, 0); break;
            case 32 * 16 + 2: AID_W8R(32, 2, 2, 0); break;
            case 32 * 16 + 4: AID_W8R(32, 4, 2, 0); break;
            default: aid_set_error("aid_conv2d: F(8,3) tile shape not instantiated"); return AID_E_BADARG;
        }
        AID_CHECK_LAUNCH();
    }
#undef AID_W8R
    if (kg) aid_note_kernel(m96 ? "conv53_wino8r_ks_kernel(64+32)" : "conv53_wino8r_ks_kernel");
    else aid_note_kernel(m96 ? "conv53_wino8r_kernel(64+32)" : (plan[0].NC == 1 ? "conv53_wino8r_kernel" : "conv53_wino8r_kernel(multi-class)"));
    return AID_OK;
}

extern "C" int aid_conv2d_wino8_supported(int Cin, int Cout, int F, int T, int dilF) {
    if (!wino_v_shape_ok(Cin, Cout, T) || (T % 32)) return 0;
    aid_conv2d_params q = {};
    q.B = 1; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.KH = 5; q.KW = 3;
    aid_conv2d_pack_dims(Cin, Cout, &q.Cin_pad, &q.Cout_pad);
    Wino4rPlan plan[2];
    return wino8r_geometry(&q, plan) ? 1 : 0;
}

extern "C" int aid_conv2d_wino_form(int B, int Cin, int Cout, int F, int T, int dilF) {
    aid_conv2d_params q = {};
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.KH = 5; q.KW = 3;
    aid_conv2d_pack_dims(Cin, Cout, &q.Cin_pad, &q.Cout_pad);
    const int f = wino_form_choice(&q);
    if (f == 4 && !aid_conv2d_wino_input_ok(B, Cin, Cout, F, T, dilF)) return 0;
    return f;
}

extern "C" int aid_conv2d_fin_supported(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino) {
    if (x_wino == 3 || x_wino == 4) return (B >= 1 && (Cout % 8) == 0 && aid_conv2d_wino2d_supported(Cin, Cout, F, T, dilF) && (x_wino == 3 || T % 32 == 0)) ? 1 : 0;      // (the LAST block of the output pass folds)
    if (B < 1 || (Cout % 8) || !wino_v_shape_ok(Cin, Cout, T)) return 0;
    if (x_wino == 2) return aid_conv2d_wino8_supported(Cin, Cout, F, T, dilF);
    if (x_wino != 1) return 0;
    aid_conv2d_params q = {};
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.KH = 5; q.KW = 3;
    aid_conv2d_pack_dims(Cin, Cout, &q.Cin_pad, &q.Cout_pad);
    Wino4rPlan plan[2];
    return wino4r_geometry(&q, plan) ? 1 : 0;
}

// scratch of the split-K instances (aid_kernels.h: ws): 0 when a launch of this shape is not split
extern "C" int64_t aid_conv2d_wino_split_ws_bytes(int B, int Cin, int Cout, int F, int T, int dilF) {
    if (!wino_v_shape_ok(Cin, Cout, T)) return 0;
    aid_conv2d_params q = {};
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.KH = 5; q.KW = 3;
    aid_conv2d_pack_dims(Cin, Cout, &q.Cin_pad, &q.Cout_pad);
    Wino4rPlan plan[2];
    const int nl = wino4r_geometry(&q, plan);
    if (nl != 1) return 0;
    const int64_t ntiles = (int64_t)B * (dilF / plan[0].NC) * plan[0].quads * plan[0].ttiles * (q.Cout_pad / 64);
    if (wino4r_ks_wanted(&q, nl, plan, ntiles)) return 0;       // (the K-group instance takes the launch: no scratch)
    return wino4r_splits(&q, nl, ntiles) == 2 ? aid_w4r_split_bytes(ntiles) : 0;
}

// Output positions per tile of the 64-wide F(4,3) kernels: 512, or 256 when a launch would otherwise put fewer than ~0.75
// workgroups on each CU (small batches: the deep levels have few positions) -- twice the workgroups, 4 waves each.
static int wino_tile_n(int B, int Cout_pad, int F, int T) {
    if (Cout_pad % 64) return 512;
    int TT = aid_pow2ceil(T); if (TT > 512) TT = 512;
    const int64_t tiles = (int64_t)aid_cdiv((int64_t)B * F, 512 / TT) * aid_cdiv(T, TT) * (Cout_pad / 64);
    return (tiles <= 160 && aid_pow2ceil(T) >= 16) ? 256 : 512;     // measured: B=1 +12 %, B=2 neutral, B>=4 never triggers on the shipped networks
}

static bool wino_v_shape_ok(int Cin, int Cout, int T) {
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    return (Cin % 4) == 0 && Cout >= 64 && ((cop % 64) == 0 || (cop % 96) == 0) && (T % 16) == 0 && T >= 16;
}

extern "C" int aid_conv2d_wino_input_supported(int Cin, int Cout, int T) { return (wino_v_shape_ok(Cin, Cout, T) && T >= 32) ? 1 : 0; }

// Geometry-aware form of aid_conv2d_wino_input_supported: T = 16 layers have only the row-shared kernel (the 512-position tiles of the fallback
// kernels would span 32 rows), so there the answer also depends on F and the dilation.
extern "C" int aid_conv2d_wino_input_ok(int B, int Cin, int Cout, int F, int T, int dilF) {
    if (!wino_v_shape_ok(Cin, Cout, T)) return 0;
    if (T >= 32) return 1;
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    aid_conv2d_params q{};
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.Cout_pad = cop;
    Wino4rPlan plan[2];
    return wino4r_geometry(&q, plan) == 1 ? 1 : 0;
}

// partials per (sample, group) of the forward (sum, sum of squares) option: the row-shared kernel only
int aid_w2d_partials(int Cout, int F, int T, int dilF, int TF);      // aid_wino2d.hip

extern "C" int aid_conv2d_stat_partials(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino) {
    if (x_wino == 3 || x_wino == 4) return (aid_conv2d_wino2d_supported(Cin, Cout, F, T, dilF) && (x_wino == 3 || T % 32 == 0)) ? aid_w2d_partials(Cout, F, T, dilF, x_wino == 4 ? 8 : 4) : 0;
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    if (!x_wino || (Cout % 8) || !wino_v_shape_ok(Cin, Cout, T)) return 0;
    const int mblk = (cop % 64 == 0) ? 64 : ((cop % 96 == 0) ? 96 : 0);
    const int cpg = Cout / 8;
    if (!mblk || (cpg % 4) || (mblk % cpg)) return 0;
    aid_conv2d_params q{};
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.Cout_pad = cop;
    Wino4rPlan plan[2];
    const int nl = x_wino == 2 ? wino8r_geometry(&q, plan) : wino4r_geometry(&q, plan);
    int n = 0;
    for (int l = 0; l < nl; ++l) n += (dilF / plan[l].NC) * plan[l].quads * plan[l].ttiles;
    return n;
}

// tiles per sample of the F(4,3) kernels (64|96 x 512 tiles) when the per-tile <y, aux> partials are well defined
extern "C" int aid_conv2d_dot_partials(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino) {
    if (x_wino == 3 || x_wino == 4) return (aid_conv2d_wino2d_supported(Cin, Cout, F, T, dilF) && (x_wino == 3 || T % 32 == 0)) ? aid_w2d_partials(Cout, F, T, dilF, x_wino == 4 ? 8 : 4) : 0;
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    if ((Cin % 4) || Cout < 64 || (Cout % 8) || (T % 4) || aid_pow2ceil(T) < 8) return 0;
    const int mblk = (cop % 64 == 0) ? 64 : ((cop % 96 == 0) ? 96 : 0);
    const int cpg = Cout / 8;
    if (!mblk || (cpg % 4) || (mblk % cpg)) return 0;
    if (x_wino && wino_v_shape_ok(Cin, Cout, T)) {           // the row-shared kernel takes the launch when its geometry fits
        aid_conv2d_params q{};
        q.B = B; q.Cin = Cin; q.Cout = Cout; q.F = F; q.T = T; q.dilF = dilF; q.Cout_pad = cop;
        Wino4rPlan plan[2];
        const int nl = x_wino == 2 ? wino8r_geometry(&q, plan) : wino4r_geometry(&q, plan);
        int n = 0;
        for (int l = 0; l < nl; ++l) n += (dilF / plan[l].NC) * plan[l].quads * plan[l].ttiles;
        if (nl || x_wino == 2) return n;
    }
    const int nblk = wino_tile_n(B, cop, F, T);
    int TT = aid_pow2ceil(T);
    if (TT > nblk) TT = nblk;
    const int ROWS = nblk / TT;
    if (ROWS > 16 || (F % ROWS)) return 0;
    return (F / ROWS) * aid_cdiv(T, TT);
}

// x_wino = 1: the caller already wrote the Winograd-domain input; there is no other kernel that can read it
static int conv53_wino_v(const aid_conv2d_params* p, hipStream_t st) {
    auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
    AID_REQUIRE((int64_t)p->Cin * p->x.sC * 4 < (1LL << 32) && (int64_t)p->wino_taps * p->Cin_pad * p->Cout_pad * 4 < (1LL << 32),
                "aid_conv2d: x_wino kernels address one sample and the weight pack with 32-bit byte offsets");
    if (p->x_wino == 2) {                                   // F(8,3): [B, Cin, F, 10, T/8] input, 50-tap pack
        AID_REQUIRE(p->wp_wino && p->wino_taps == 50 && p->KH == 5 && p->KW == 3 && !p->in_scale && p->act == 0,
                    "aid_conv2d: x_wino = 2 needs the 50-tap F(8,3) pack of a 5x3 layer and no in-kernel prologue");
        AID_REQUIRE(wino_v_shape_ok(p->Cin, p->Cout, p->T) && (p->T % 32) == 0, "aid_conv2d: shape not supported with x_wino = 2 (see aid_conv2d_wino_form)");
        AID_REQUIRE(p->x.sF >= 10 * (p->T / 8) && (int64_t)4 * p->x.sC < (1LL << 31), "aid_conv2d: x_wino = 2 rows are [10][T/8]");
        AID_REQUIRE(al4(p->x) && al4(p->y) && (!p->res.p || al4(p->res)) && (!p->aux.p || al4(p->aux)), "aid_conv2d: x_wino views must be 16-byte aligned");
        return launch_wino8r(p, st);
    }
    AID_REQUIRE(p->wp_wino && p->wino_taps == 30 && p->KH == 5 && p->KW == 3 && !p->in_scale && p->act == 0,
                "aid_conv2d: x_wino needs the 30-tap Winograd pack of a 5x3 layer and no in-kernel prologue");
    AID_REQUIRE(wino_v_shape_ok(p->Cin, p->Cout, p->T), "aid_conv2d: shape not supported with x_wino (see aid_conv2d_wino_input_supported)");
    AID_REQUIRE(p->x.sF >= 6 * (p->T / 4) && (int64_t)4 * p->x.sC < (1LL << 31), "aid_conv2d: x_wino rows are [6][T/4]");
    AID_REQUIRE(al4(p->x) && al4(p->y) && (!p->res.p || al4(p->res)) && (!p->aux.p || al4(p->aux)), "aid_conv2d: x_wino views must be 16-byte aligned");
    int rc;
    rc = launch_wino4r(p, st);                              // row-shared tiles, two workgroups per CU (when the geometry fits)
    if (rc != 1000) return rc;
    AID_REQUIRE(!p->fin_mode, "aid_conv2d: fin_mode needs the row-shared kernel (aid_conv2d_fin_supported)");
    if (p->Cout_pad % 64 == 0 && wino_tile_n(p->B, p->Cout_pad, p->F, p->T) == 256)
                               rc = launch_wino4v<1, 1, 2, 2, 16, 2, 1>(p, st);     // 64 x 256, 4 waves (one per SIMD: 92 KB of LDS): small grids
    else if (p->Cout_pad % 64 == 0) rc = launch_wino4v<1, 1, 2, 4, 16, 2, 2>(p, st);     // 64 x 512, 8 waves
    else if (p->Cout_pad % 96 == 0) rc = launch_wino4v<1, 1, 3, 4, 16, 2, 3>(p, st);     // 96 x 512, 12 waves (geometries the row-shared kernel declines)
    else { aid_set_error("aid_conv2d: x_wino needs a 64- or 96-wide Cout tile"); return AID_E_BADARG; }
    AID_REQUIRE(rc != 1000, "aid_conv2d: x_wino tile does not fit this T");
    return rc;
}

// returns 1 if the Winograd kernel took the launch, 0 if not eligible, <0 on error
int aid_conv53_wino_try(const aid_conv2d_params* p, hipStream_t st) {
    if (p->x_wino) { const int rc = conv53_wino_v(p, st); return rc == AID_OK ? 1 : rc; }
    if (p->fin_mode) { aid_set_error("aid_conv2d: fin_mode needs Winograd-domain input on the row-shared kernel (aid_conv2d_fin_supported)"); return AID_E_BADARG; }
    if (!p->wp_wino || !(p->KH == 5 && p->KW == 3) || p->in_scale || p->act != 0) return 0;
    if ((p->Cin % 4) != 0 || p->Cout < 64 || (p->T % 4) != 0) return 0;
    if (aid_pow2ceil(p->T) < 8) return 0;
    if ((p->x.sB % 4) || (p->x.sC % 4) || (p->x.sF % 4) || (((uintptr_t)p->x.p) & 15)) return 0;
    if ((p->y.sB % 2) || (p->y.sC % 2) || (p->y.sF % 2) || (((uintptr_t)p->y.p) & 7)) return 0;
    if (p->res.p && ((p->res.sB % 2) || (p->res.sC % 2) || (p->res.sF % 2) || (((uintptr_t)p->res.p) & 7))) return 0;
    if (p->aux.p && ((p->aux.sB % 2) || (p->aux.sC % 2) || (p->aux.sF % 2) || (((uintptr_t)p->aux.p) & 7))) return 0;
    if ((int64_t)4 * p->x.sC >= (1LL << 31)) return 0;
    if (p->wino_taps != 30) return 0;                      // the 30-tap F(4,3) pack is the only Winograd pack
    if ((p->y.sB % 4) || (p->y.sC % 4) || (p->y.sF % 4) || (((uintptr_t)p->y.p) & 15)) return 0;
    if (p->res.p && ((p->res.sB % 4) || (p->res.sC % 4) || (p->res.sF % 4) || (((uintptr_t)p->res.p) & 15))) return 0;
    if (p->aux.p && ((p->aux.sB % 4) || (p->aux.sC % 4) || (p->aux.sF % 4) || (((uintptr_t)p->aux.p) & 15))) return 0;
    int rc;
    if (p->Cout_pad % 64 == 0 && wino_tile_n(p->B, p->Cout_pad, p->F, p->T) == 256)
                                    rc = launch_wino4<1, 1, 2, 2, 16, 2, 2>(p, st);   // 64 x 256, 4 waves: small grids
    else if (p->Cout_pad % 64 == 0) rc = launch_wino4<1, 1, 2, 4, 16, 2, 2>(p, st);   // 64 x 512, 8 waves, 1 workgroup / CU
    else if (p->Cout_pad % 96 == 0) rc = launch_wino4<1, 1, 3, 4, 16, 2, 3>(p, st);   // 96 x 512, 12 waves
    else return 0;
    if (rc == 1000) return 0;
    return rc == AID_OK ? 1 : rc;
}
