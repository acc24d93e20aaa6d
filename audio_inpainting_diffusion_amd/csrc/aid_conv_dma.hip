// conv53_dma_kernel: the 5x3 dilated dense convolution (plain-copy prologue) with DIRECT-TO-LDS staging.
//
// Same GEMM decomposition and epilogue as conv_mfma_kernel (aid_conv.hip), but for the shapes that carry the
// FLOPs (KH=5, KW=3, Cin % 4 == 0, no in-kernel prologue because aid_scale_act ran first):
//   * both operands travel HBM/L2 -> LDS with `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR
//     round trip, no ds_write, no VALU besides one 64-bit pointer bump per piece per chunk).  The LDS image
//     is therefore lane-linear: the x tile stores, per (ci, kh), the N_BLK core samples of the tile
//     CONTIGUOUSLY (position n = rr*TT + tt) followed by a small halo area [rr][left,right]; out-of-range
//     rows / columns / channels are sourced from a device zero page instead of being masked, so every lane of
//     every DMA is active and no stale LDS data survives;
//   * with the core region a compile-time constant per (ci,kh), every operand ds_read is `vaddr + immediate`:
//     the per-lane addresses (incl. the halo redirection of the tile-edge lanes for kw = 0/2) are computed once,
//     so the steady-state loop is [ds_read_b32, v_mfma] plus 3-5 DMA issues per wave per chunk;
//   * double-buffered LDS, DMA of chunk c+1 issued right after the barrier that closes chunk c-1, one
//     `__syncthreads()` (vmcnt(0) + barrier) per chunk, priority-based progress balancing between the waves
//     of a SIMD (see aid_conv.hip).
#include "aid_common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDmaDev {
    aid_conv2d_params p;
    const float* zero;   // >= 64 B of zeros in global memory
    int tt_log2, rows_log2, tiles_t, nrows, nchunks;
};

__device__ float4 g_aid_zero_page[16];

#define GLDS16(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int MT, int NT, int WGM, int WGN, int RMAX, int KC, int MINW>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv53_dma_kernel(const ConvDmaDev a) {
    constexpr int KH = 5, KW = 3, TAPS = 15;
    constexpr int NW = WGM * WGN;
    constexpr int NTHREADS = 64 * NW;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    constexpr int HALO = 2 * RMAX;                      // halo floats per (ci,kh) block: [rr < RMAX][2]
    constexpr int XB = N_BLK + HALO;                    // floats per (ci,kh) block
    constexpr int XSZ = KC * KH * XB;
    constexpr int WROW = (M_BLK % 64 == 0) ? M_BLK : ((M_BLK + 63) / 64) * 64;   // LDS weight row stride (96 -> 128)
    constexpr int WSZ_RAW = TAPS * KC * WROW;
    constexpr int WSZ = ((WSZ_RAW + 255) / 256) * 256;   // whole 1-KiB pieces (tail lanes read the zero page)
    constexpr int BUFSZ = XSZ + WSZ;
    constexpr int NXP = KC * KH * (N_BLK / 256);        // 1-KiB DMA pieces of the x tile
    constexpr int NWP = WSZ / 256;                      // 1-KiB DMA pieces of the weight tile
    constexpr int NP = NXP + NWP;
    constexpr int PPW = (NP + NW - 1) / NW;             // pieces per wave per chunk
    constexpr int HQ = (2 * KC * KH * RMAX + NTHREADS - 1) / NTHREADS;   // halo scalars per thread (ROWS <= RMAX)
    static_assert(N_BLK % 256 == 0 && WSZ % 256 == 0, "tiles must be whole 1-KiB pieces");

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* rowinfo = (int*)(smem + 2 * BUFSZ);            // [ROWS][2] = (b, f) or b = -1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN;
    const int wn = wave % WGN;

    const int tile_t = blockIdx.x % a.tiles_t;
    const int rg = blockIdx.x / a.tiles_t;
    const int row0 = rg << a.rows_log2;
    const int t0 = tile_t << a.tt_log2;
    const int m0 = blockIdx.y * M_BLK;

    for (int r = tid; r < ROWS; r += NTHREADS) {
        const int rid = row0 + r;
        int b = -1, f = 0;
        if (rid < a.nrows) { b = rid / p.F; f = rid - b * p.F; }
        rowinfo[2 * r] = b;
        rowinfo[2 * r + 1] = f;
    }
    __syncthreads();

    // ---- DMA piece descriptors (per lane, constant over the K loop) ----------------------------------------
    const float* psrc[PPW];     // source of this lane's 16 bytes for chunk 0 (or the zero page)
    int pstride[PPW];           // element advance per chunk (0 for the zero page)
    int plds[PPW];              // wave-uniform LDS float offset of the piece inside a buffer, -1 = no piece
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        psrc[i] = a.zero; pstride[i] = 0; plds[i] = -1;
        if (pc < NXP) {
            const int blk = pc / (N_BLK / 256);         // ci*KH + kh
            const int sub = pc % (N_BLK / 256);
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
            const int wp_ = pc - NXP;                   // piece index inside the weight tile
            const int e = wp_ * 256 + 4 * lane;         // float offset inside the weight tile [tap][ci][WROW]
            const int row = e / WROW, col = e % WROW;   // row = tap*KC + ci
            const int tap = row / KC, ci = row % KC;
            plds[i] = XSZ + wp_ * 256;
            if (col < M_BLK && e < WSZ_RAW) {            // (columns 96..127 of the 96-row tile, and the tail, stay zero)
                psrc[i] = p.wp + ((int64_t)tap * p.Cin_pad + ci) * p.Cout_pad + m0 + col;
                pstride[i] = KC * p.Cout_pad;
            }
        }
    }
    // ---- halo descriptors -----------------------------------------------------------------------------------
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
            const int blk = (h >> 1) >> a.rows_log2;    // ci*KH + kh
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
    // ---- operand addresses (LDS float offsets inside a buffer; ci / kh / tap terms are immediates) -----------
    const int half = lane >> 5;                          // which of the two ci of a k-step this lane feeds
    int vB[NT][KW];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + (lane & 31);
        const int rr = n >> a.tt_log2, tt = n & (TT - 1);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const int tc = tt + kw - 1;
            int off = n + kw - 1;
            if (tc < 0) off = N_BLK + rr * 2;
            else if (tc >= TT) off = N_BLK + rr * 2 + 1;
            vB[j][kw] = half * KH * XB + off;
        }
    }
    int vA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) vA[i] = XSZ + half * WROW + (wm * MT + i) * 32 + (lane & 31);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float hv[HQ];
    auto issue_dma = [&](int ch, float* buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (plds[i] >= 0) {                          // wave-uniform
                const float* src = psrc[i] + (int64_t)ch * pstride[i];
                GLDS16(src, buf + plds[i]);
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

    issue_dma(0, smem);
    write_halo(smem);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = smem + cur * BUFSZ;
        float* Nx = smem + (cur ^ 1) * BUFSZ;
        const bool more = (ch + 1) < a.nchunks;
        if (more) issue_dma(ch + 1, Nx);
        float av[2][MT], bv[2][NT];
        auto load_frags = [&](int ks, int buf) {
            const int tap = ks / (KC / 2), cp = ks % (KC / 2);
            const int kh = tap / KW, kw = tap % KW;
#pragma unroll
            for (int i = 0; i < MT; ++i) av[buf][i] = Bf[vA[i] + (tap * KC + 2 * cp) * WROW];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[buf][j] = Bf[vB[j][kw] + (2 * cp * KH + kh) * XB];
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < TAPS * (KC / 2); ++ks) {
            if (ks == 0) __builtin_amdgcn_s_setprio(3);
            else if (ks == 3 * (KC / 2)) __builtin_amdgcn_s_setprio(2);
            else if (ks == 6 * (KC / 2)) __builtin_amdgcn_s_setprio(1);
            else if (ks == 9 * (KC / 2)) __builtin_amdgcn_s_setprio(0);
            if (ks + 1 < TAPS * (KC / 2)) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
        }
        if (more) write_halo(Nx);
        __syncthreads();
    };
    for (int ch = 0; ch < a.nchunks; ch += 2) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < a.nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue (identical to conv_mfma_kernel) ---------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + (lane & 31);
        const int rr = n >> a.tt_log2;
        const int tt = n & (TT - 1);
        const int b = rowinfo[2 * rr];
        const int f = rowinfo[2 * rr + 1];
        const int t = t0 + tt;
        if (b < 0 || t >= p.T) continue;
        const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
        const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
        const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // gather phase first (res / aux / gate), then compute + store: res may alias y (gradient accumulation),
            // so the compiler would otherwise serialise load -> store element by element (16 dependent round trips)
            float rv[16], uv[16], sv[16];
            const int mbase = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                const bool ok = m < p.Cout;
                rv[r] = (ok && p.res.p) ? p.res.p[rbase + (int64_t)m * p.res.sC] : 0.f;
                sv[r] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                uv[r] = (ok && p.epi == 1) ? p.aux.p[abase + (int64_t)m * p.aux.sC] * p.aux_scale[(int64_t)b * p.aux_scale_ld + m] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m >= p.Cout) continue;
                float v = acc[i][j][r] * sv[r];
                if (p.epi == 1) v *= aid_dgelu(uv[r]);
                v += p.res_scale * rv[r];
                p.y.p[ybase + (int64_t)m * p.y.sC] = p.alpha * v;
            }
        }
    }
}

template <int MT, int NT, int WGM, int WGN, int RMAX, int KC = 4, int MINW = 1>
static int launch_dma(const aid_conv2d_params* p, hipStream_t st) {
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    constexpr int WROW = (M_BLK % 64 == 0) ? M_BLK : ((M_BLK + 63) / 64) * 64;
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    ConvDmaDev a;
    a.p = *p;
    a.zero = zero;
    int TT = aid_pow2ceil(p->T);
    if (TT > N_BLK) TT = N_BLK;
    a.tt_log2 = aid_ilog2(TT);
    const int ROWS = N_BLK / TT;
    a.rows_log2 = aid_ilog2(ROWS);
    if (ROWS > RMAX) return 1000;                          // halo area holds RMAX rows: caller falls back
    a.tiles_t = aid_cdiv(p->T, TT);
    a.nrows = p->B * p->F;
    a.nchunks = p->Cin / KC;
    const int rgroups = aid_cdiv(a.nrows, ROWS);
    dim3 grid((unsigned)(rgroups * a.tiles_t), (unsigned)(p->Cout_pad / M_BLK));
    const size_t lds = sizeof(float) * 2 * ((size_t)KC * 5 * (N_BLK + 2 * RMAX) + (size_t)((15 * KC * WROW + 255) / 256) * 256) + sizeof(int) * 2 * ROWS;
    auto kern = conv53_dma_kernel<MT, NT, WGM, WGN, RMAX, KC, MINW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel("conv53_dma_kernel");
    return AID_OK;
}

// returns 1 if the DMA kernel took the launch, 0 if the shape is not eligible, <0 on error
int aid_conv53_dma_try(const aid_conv2d_params* p, hipStream_t st) {
    if (!(p->KH == 5 && p->KW == 3) || p->in_scale || p->act != 0) return 0;
    if ((p->Cin % 4) != 0 || p->Cout < 64 || (p->T % 4) != 0) return 0;
    const int TT = aid_pow2ceil(p->T);
    if (TT < 8) return 0;                                  // ROWS <= 32 (halo area)
    if ((p->x.sB % 4) || (p->x.sC % 4) || (p->x.sF % 4) || (((uintptr_t)p->x.p) & 15)) return 0;
    if ((int64_t)4 * p->x.sC >= (1LL << 31)) return 0;
    int rc;
    // (round-1 experiments with 4/16-wave tiles, KC = 2 and 3-4 workgroups per CU are recorded in DESIGN.md section 3.1)
    if (p->Cout_pad % 64 == 0)       rc = launch_dma<1, 2, 2, 4, 16>(p, st);     //  64 x 256, 8 waves, 2 workgroups per CU
    else if (p->Cout_pad % 96 == 0)  rc = launch_dma<1, 2, 3, 4, 32>(p, st);     //  96 x 256, 12 waves
    else return 0;
    if (rc == 1000) return 0;
    return rc == AID_OK ? 1 : rc;
}
