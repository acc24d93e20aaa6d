// conv11_dma_kernel: 1x1 channel projections (res_conv / skip projections, unet...py:412-415,:488-491) as an
// fp32-MFMA GEMM with DIRECT-TO-LDS staging and small workgroup footprints.
//
// Why a third 1x1 kernel: these layers sit at the ridge (40-130 FLOP/B).  The register-staged tiled kernel
// (conv_mfma_kernel) runs a tile's load, MFMA and store phases back to back with one workgroup per CU; the streaming kernel
// (aid_conv1x1.hip) never overlaps a wave's loads, MFMAs and stores either.  Measured 12-20 % ahead of both.  Here
//   * both operands go HBM/L2 -> LDS with global_load_lds_dwordx4 into two STATIC buffers (see aid_conv_wino.hip: distinct
//     LDS objects keep the loads of chunk c+1 in flight while chunk c is multiplied);
//   * a buffer is 24 KB (16 channels x 256 positions + 16 x 128 weights) and a wave holds 64 accumulators, so two to three
//     workgroups share a CU: one tile's store-heavy epilogue overlaps another tile's K loop;
//   * a lane's NT = 2 position tiles are INTERLEAVED (tile j = positions 2n + j), so the B fragments of both tiles are ONE
//     conflict-free ds_read_b64 and the epilogue stores float2.
// Per-(b,ci) prologue scale (in_scale) is applied to the A fragment (a tile never straddles two samples: host-checked).
#include "aid_common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
int aid_conv_splitk_reduce(const aid_conv2d_params* p, const float* ws, int splits, hipStream_t st);      // aid_conv.hip

struct C11Dev {
    aid_conv2d_params p;
    const float* zero;
    int tt_log2, rows_log2, tiles_t, nrows, nchunks;
    int nx, ny, per_xcd;
    int nch1;                  // chunks served by p.x (all of them unless p.x2 is given: then chunks >= nch1 read channels of p.x2)
    int splits, cps;           // split-K (grid-starved GEMMs: the qk projections, N = B*T columns only): blockIdx.y walks `splits` ranges of
    float* ws;                 // `cps` chunks; raw partial sums go to ws[z][B][Cout][F][T], reduced in a fixed order by aid_conv_splitk_reduce
};

__device__ float4 g_aid_zero_page_c11[16];

#define GLDS16C(gptr, lptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// MT m-tiles (32 cout) x 2 interleaved position tiles per wave; WGM x WGN waves; N_BLK = 64 * WGN positions
template <int MT, int WGM, int WGN, int RMAX, int KC, int MINW>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv11_dma_kernel(const C11Dev a) {
    constexpr int NT = 2;
    constexpr int NW = WGM * WGN;
    constexpr int NTHREADS = 64 * NW;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    constexpr int XSZ = KC * N_BLK;
    constexpr int WROW = (M_BLK % 64 == 0) ? M_BLK : ((M_BLK + 63) / 64) * 64;
    constexpr int WSZ = KC * WROW;
    constexpr int BUFSZ = XSZ + WSZ;
    constexpr int NXP = XSZ / 256;
    constexpr int NWP = WSZ / 256;
    constexpr int NP = NXP + NWP;
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int NSTEP = KC / 2;
    static_assert(N_BLK % 256 == 0 && WSZ % 256 == 0, "whole 1-KiB pieces");

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;

    __shared__ __attribute__((aligned(16))) float sbuf0[BUFSZ];
    __shared__ __attribute__((aligned(16))) float sbuf1[BUFSZ];
    __shared__ int rowinfo[2 * RMAX];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN;
    const int wn = wave % WGN;

    // XCD-aware tile order (see conv53_wino4_kernel): one contiguous (row-group, Cout-tile) range per XCD, Cout tile fastest
    const int Lt = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (Lt >= a.nx * a.ny) return;
    const int bx = Lt / a.ny;
    const int by = Lt - bx * a.ny;
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

    // ---- DMA piece descriptors ---------------------------------------------------------------------------------------
    const float* psrc[PPW];
    const float* psrc2[PPW];                             // second K segment (p.x2): same piece, channels of the other tensor
    int pstride[PPW], pstride2[PPW], plds[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + i * NW;
        psrc[i] = a.zero; pstride[i] = 0; plds[i] = -1;
        psrc2[i] = a.zero; pstride2[i] = 0;
        if (pc < NXP) {
            const int ci = pc / (N_BLK / 256), sub = pc % (N_BLK / 256);
            const int n = sub * 256 + 4 * lane;
            const int rr = n >> a.tt_log2, tt = n & (TT - 1);
            const int b = rowinfo[2 * rr];
            const int f = rowinfo[2 * rr + 1];
            plds[i] = ci * N_BLK + sub * 256;
            if (b >= 0 && t0 + tt < p.T) {
                psrc[i] = p.x.p + (int64_t)b * p.x.sB + (int64_t)ci * p.x.sC + (int64_t)f * p.x.sF + t0 + tt;
                pstride[i] = (int)(KC * p.x.sC);
                if (p.x2.p) {
                    psrc2[i] = p.x2.p + (int64_t)b * p.x2.sB + (int64_t)ci * p.x2.sC + (int64_t)f * p.x2.sF + t0 + tt;
                    pstride2[i] = (int)(KC * p.x2.sC);
                }
            }
        } else if (pc < NP) {
            const int wp_ = pc - NXP;
            const int e = wp_ * 256 + 4 * lane;
            const int ci = e / WROW, col = e % WROW;
            plds[i] = XSZ + wp_ * 256;
            if (col < M_BLK) {
                psrc[i] = p.wp + (int64_t)ci * p.Cout_pad + m0 + col;
                pstride[i] = KC * p.Cout_pad;
                psrc2[i] = psrc[i] + (int64_t)a.nch1 * pstride[i];       // (weights: one stacked pack, K continues)
                pstride2[i] = pstride[i];
            }
        }
    }
    // ---- operand addresses ---------------------------------------------------------------------------------------------
    const int half = lane >> 5;
    const int vB = half * N_BLK + wn * 64 + 2 * (lane & 31);        // float2: positions (2n, 2n+1) of this wave's 64
    int vA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) vA[i] = XSZ + half * WROW + (wm * MT + i) * 32 + (lane & 31);
    const int b_tile = rowinfo[0];
    const float* sp = (p.in_scale && b_tile >= 0) ? p.in_scale + (int64_t)b_tile * p.in_scale_ld + half : nullptr;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_dma = [&](int ch, float* buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (plds[i] >= 0) {
                const float* src = ch < a.nch1 ? psrc[i] + (int64_t)ch * pstride[i] : psrc2[i] + (int64_t)(ch - a.nch1) * pstride2[i];
                GLDS16C(src, buf + plds[i]);
            }
        }
    };
    float sc[2][NSTEP];                                              // prologue scales of the current / next chunk
    auto load_scales = [&](int ch, int q) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int ci = ch * KC + 2 * s + half;
            sc[q][s] = (sp && ci < p.Cin) ? sp[ch * KC + 2 * s] : 1.f;
        }
    };

    const int ch_lo = blockIdx.y * a.cps;                            // this workgroup's K range (the whole K unless split)
    const int ch_hi = min(a.nchunks, ch_lo + a.cps);
    issue_dma(ch_lo, sbuf0);
    load_scales(ch_lo, 0);
    __syncthreads();

    auto chunk = [&](auto curc, int ch) {
        constexpr int cur = decltype(curc)::value;
        const float* Bf = cur ? sbuf1 : sbuf0;
        float* Nx = cur ? sbuf0 : sbuf1;
        const bool more = (ch + 1) < ch_hi;
        if (more) { issue_dma(ch + 1, Nx); load_scales(ch + 1, cur ^ 1); }
        float2 bv[2];
        float av[2][MT];
        auto load_step = [&](int s, int buf) {
            bv[buf] = *reinterpret_cast<const float2*>(Bf + vB + 2 * s * N_BLK);
#pragma unroll
            for (int i = 0; i < MT; ++i) av[buf][i] = Bf[vA[i] + 2 * s * WROW];
        };
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
            const int bq = s & 1;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float aw = av[bq][i] * sc[cur][s];
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bv[bq].x, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bv[bq].y, acc[i][1], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    for (int ch = ch_lo; ch < ch_hi; ch += 2) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < ch_hi) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }

    // ---- epilogue: two consecutive positions per lane and output channel ---------------------------------------------
    const int n = wn * 64 + 2 * (lane & 31);
    const int rr = n >> a.tt_log2, tt = n & (TT - 1);
    const int b = rowinfo[2 * rr];
    const int f = rowinfo[2 * rr + 1];
    const int t = t0 + tt;
    const bool live = !(b < 0 || t >= p.T);                          // T % 2 == 0: both samples in range together
    if (!live && !p.dot_ws) return;
    if (live && a.splits > 1) {                                      // split-K: raw partial sums (float2 per output channel)
        const int64_t ft = (int64_t)p.F * p.T;
        float* wsb = a.ws + (((int64_t)blockIdx.y * p.B + b) * p.Cout) * ft + (int64_t)f * p.T + t;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = m0 + (wm * MT + i) * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < p.Cout) *reinterpret_cast<float2*>(wsb + (int64_t)m * ft) = make_float2(acc[i][0][r], acc[i][1][r]);
            }
        }
    }
    if (a.splits > 1) return;
    const int64_t ybase = (int64_t)b * p.y.sB + (int64_t)f * p.y.sF + t;
    const int64_t rbase = p.res.p ? ((int64_t)b * p.res.sB + (int64_t)f * p.res.sF + t) : 0;
    const int64_t abase = p.aux.p ? ((int64_t)b * p.aux.sB + (int64_t)f * p.aux.sF + t) : 0;
    float dsum[MT][4];                                               // <y, aux> per block of 4 rows (dot_ws)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) dsum[i][q] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (!live) break;
        const int mbase = m0 + (wm * MT + i) * 32 + 4 * half;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {             // gather 4 rows, then compute + store (res may alias y)
            float2 rv[4], uv[4];
            float sv[4], as[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                const bool ok = m < p.Cout;
                rv[q] = (ok && p.res.p) ? *reinterpret_cast<const float2*>(p.res.p + rbase + (int64_t)m * p.res.sC) : make_float2(0.f, 0.f);
                sv[q] = (ok && p.out_scale) ? p.out_scale[(int64_t)b * p.out_scale_ld + m] : 1.f;
                if (ok && p.epi == 1) {
                    as[q] = p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                    uv[q] = *reinterpret_cast<const float2*>(p.aux.p + abase + (int64_t)m * p.aux.sC);
                } else { as[q] = 0.f; uv[q] = make_float2(0.f, 0.f); }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m >= p.Cout) continue;
                float y0 = acc[i][0][r] * sv[q], y1 = acc[i][1][r] * sv[q];
                if (p.epi == 1) { y0 *= aid_dgelu(uv[q].x * as[q]); y1 *= aid_dgelu(uv[q].y * as[q]); }
                y0 += p.res_scale * rv[q].x;
                y1 += p.res_scale * rv[q].y;
                y0 *= p.alpha; y1 *= p.alpha;
                *reinterpret_cast<float2*>(p.y.p + ybase + (int64_t)m * p.y.sC) = make_float2(y0, y1);
                if (p.dot_ws) dsum[i][r0 >> 2] += y0 * uv[q].x + y1 * uv[q].y;
            }
        }
    }
    if (p.dot_ws) {                                                  // per (sample, channel group): one partial per position tile (as conv53_wino4r_kernel)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = live ? dsum[i][q] : 0.f;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);
                dsum[i][q] = v;
            }
        float* red = sbuf0;                                          // (every wave is past the last chunk's barrier)
        if ((lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((wave * 2 + half) * MT + i) * 4 + q] = dsum[i][q];
        }
        __syncthreads();
        const int cpg = p.Cout >> 3;                                 // channels per group (8 groups); M_BLK % cpg == 0 and cpg % 4 == 0 (host-checked)
        const int grp = m0 / cpg + tid;
        if (tid < M_BLK / cpg && grp < 8) {
            float sacc = 0.f;
            for (int w = 0; w < NW; ++w)                             // fixed order: deterministic
                for (int h = 0; h < 2; ++h)
                    for (int i = 0; i < MT; ++i)
                        for (int q = 0; q < 4; ++q) {
                            const int mrow = m0 + ((w / WGN) * MT + i) * 32 + 4 * h + 8 * q;
                            if (mrow < p.Cout && mrow / cpg == grp) sacc += red[((w * 2 + h) * MT + i) * 4 + q];
                        }
            const int bt = rowinfo[0];
            const int ptile = ((row0 - bt * p.F) >> a.rows_log2) * a.tiles_t + tile_t;
            p.dot_ws[((int64_t)bt * 8 + grp) * p.dot_n + ptile] = (double)sacc;
        }
    }
}

template <int MT, int WGM, int WGN, int RMAX, int KC, int MINW>
static int launch_c11(const aid_conv2d_params* p, hipStream_t st, int splits = 1, int kps = 0) {      // splits > 1: the partition of gemm_k_partition
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 64 * WGN;
    static const float* zero = nullptr;
    if (!zero) {
        void* z = nullptr;
        if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_aid_zero_page_c11)) != hipSuccess) { aid_set_error("aid_conv2d: zero page lookup failed"); return AID_E_LAUNCH; }
        zero = (const float*)z;
    }
    C11Dev a;
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
    a.nchunks = aid_cdiv(p->Cin, KC);
    a.nch1 = p->x2.p ? p->Cin1 / KC : a.nchunks;
    a.nx = aid_cdiv(a.nrows, ROWS) * a.tiles_t;
    a.ny = p->Cout_pad / M_BLK;
    a.per_xcd = (a.nx * a.ny + 7) / 8;
    a.cps = splits > 1 ? kps / KC : a.nchunks;
    a.splits = aid_cdiv(a.nchunks, a.cps);
    a.ws = (float*)p->ws;
    hipLaunchKernelGGL((conv11_dma_kernel<MT, WGM, WGN, RMAX, KC, MINW>), dim3((unsigned)(8 * a.per_xcd), (unsigned)a.splits), dim3(64 * WGM * WGN), 0, st, a);
    AID_CHECK_LAUNCH();
    aid_note_kernel(a.splits > 1 ? "conv11_dma_kernel+splitk" : "conv11_dma_kernel");
    if (a.splits > 1) return aid_conv_splitk_reduce(p, (const float*)p->ws, a.splits, st);
    return AID_OK;
}

// ---- skinny GEMM: the qk projections when N = B*T <= 128 columns (the reference's testers run B = 1) -------------------------------------
// y[M, N] = W^T x with K, M in the thousands and N of 32 ... 128: the weights (52 ... 103 MB per layer) are read once and nothing else matters.
// The tiled kernels stage one 8-KB weight chunk per workgroup at a time: 0.7 TB/s (latency-bound: Little's law wants ~48 KB in flight per CU).
// Here a wave owns 32 output rows x all N columns of one K range and loads BOTH operands straight from global memory into the MFMA fragments
// (weights: 2 x 128 contiguous bytes per load, x: L2-resident), U = 8 k-pairs unrolled = 16 independent loads in flight per wave, no LDS, no
// barriers; K is split over blockIdx.y and the partial sums go through the same fixed-order reduction as the other split-K paths.
template <int NT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const aid_conv2d_params p, float* __restrict__ ws, int kps) {
    constexpr int U = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= p.Cout_pad) return;
    const int half = lane >> 5, l31 = lane & 31;
    const int k0 = blockIdx.y * kps, k1 = min(p.Cin, k0 + kps);
    const float* wp = p.wp + m0 + l31;
    const float* xq[NT];
    bool ok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + l31;
        const int b = n / p.T, t = n - b * p.T;
        ok[j] = b < p.B;
        xq[j] = p.x.p + (ok[j] ? ((int64_t)b * p.x.sB + t) : 0);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kk = k0; kk < k1; kk += 2 * U) {
        float a[U], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kk + 2 * u + half;
            const bool kin = k < k1;
            a[u] = kin ? wp[(int64_t)k * p.Cout_pad] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[u][j] = (kin && ok[j]) ? xq[j][(int64_t)k * p.x.sC] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u][j], acc[j], 0, 0, 0);
    }
    const int64_t ft = p.T;                                         // F = 1
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + l31;
        const int b = n / p.T, t = n - b * p.T;
        if (b >= p.B) continue;
        float* wsb = ws + (((int64_t)blockIdx.y * p.B + b) * p.Cout) * ft + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 4 * half + (r & 3) + 8 * (r >> 2);
            if (m < p.Cout) wsb[(int64_t)m * ft] = acc[j][r];
        }
    }
}

// partials per (sample, group) of the <y, aux> option on this kernel (epi = 1 on a 1x1 layer): position tiles per sample; 0 = not available
int aid_conv1x1_rs_dot_partials(int Cin, int Cout, int cop, int F, int T);   // aid_conv1x1_rs.hip
int aid_conv1x1_rs_shape_ok(int Cin, int Cout, int cop, int F, int T);
int aid_conv1x1_rs_enabled(void);

extern "C" int aid_conv2d_dot_partials_1x1(int B, int Cin, int Cout, int F, int T) {
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    if (aid_conv1x1_rs_enabled()) {                                   // the register-streamed kernel takes the shapes it serves
        const int n = aid_conv1x1_rs_dot_partials(Cin, Cout, cop, F, T);
        if (n) return n;
        if (aid_conv1x1_rs_shape_ok(Cin, Cout, cop, F, T)) return 0;
    }
    if (F <= 1 || Cin < 32 || (Cin % 16) || Cout < 32 || (Cout % 8) || (T % 4) || aid_pow2ceil(T) < 8) return 0;
    const int mblk = (cop % 128 == 0) ? 128 : ((cop % 64 == 0) ? 64 : ((cop % 96 == 0) ? 96 : 0));
    const int cpg = Cout / 8;
    if (!mblk || (cpg % 4) || (mblk % cpg)) return 0;
    int TT = aid_pow2ceil(T); if (TT > 256) TT = 256;
    const int ROWS = 256 / TT;
    if (ROWS > 16 || (F % ROWS)) return 0;
    return (F / ROWS) * aid_cdiv(T, TT);
}

// K partition of the qk GEMMs: a function of K ALONE -- the skinny kernel (N <= 128) and the tiled kernel (any N) then add the same k-pairs in
// the same order into the same number of partial planes, so a segment's result does not depend on the batch it is evaluated in.
static void gemm_k_partition(int Cin, int* S, int* kps) {
    int s = 8;
    if (s > Cin / 64) s = Cin / 64;
    if (s < 1) s = 1;
    int cps = aid_cdiv(aid_cdiv(Cin, 16), s);          // 16-channel chunks per range, even (two LDS buffers alternate from the first chunk)
    cps += cps & 1;
    *kps = cps * 16;
    *S = aid_cdiv(Cin, *kps);
}

// shapes on which the two-tensor K axis (x2 / Cin1) is available: what aid_conv1x1_dma_try accepts below, minus the pointer alignment checks
extern "C" int aid_conv2d_x2_supported(int Cin, int Cin1, int Cout, int F, int T) {
    int cip, cop;
    aid_conv2d_pack_dims(Cin, Cout, &cip, &cop);
    if (F <= 1 || Cin < 32 || (Cin % 16) || Cout < 32 || Cin1 <= 0 || Cin1 >= Cin || (Cin1 % 16)) return 0;
    if ((T % 4) || aid_pow2ceil(T) < 8) return 0;
    int TT = aid_pow2ceil(T); if (TT > 256) TT = 256;
    if (256 / TT > 16) return 0;
    return (cop % 128 == 0 || cop % 64 == 0 || cop % 96 == 0) ? 1 : 0;
}

// returns 1 if this kernel took the launch, 0 if not eligible, <0 on error
int aid_conv1x1_dma_try(const aid_conv2d_params* p, hipStream_t st) {
    if (!(p->KH == 1 && p->KW == 1) || p->act != 0) return 0;
    if (p->Cin < 32 || (p->Cin % 16) || p->Cout < 32) return 0;      // (K is walked in whole 16-channel chunks: no ragged tail)
    const bool gemm = p->F == 1;                                     // the qk projections: a plain GEMM with N = B*T columns
    if (gemm && (p->epi != 0 || p->in_scale || p->x2.p || (p->Cout_pad % 128) || p->Cin < 256)) return 0;
    if ((p->T % 4) || aid_pow2ceil(p->T) < 8) return 0;
    auto al = [](const aid_view& v, int q) { return (v.sB % q) == 0 && (v.sC % q) == 0 && (v.sF % q) == 0 && (((uintptr_t)v.p) & (4 * q - 1)) == 0; };
    if (!al(p->x, 4) || !al(p->y, 2) || (p->res.p && !al(p->res, 2)) || (p->aux.p && !al(p->aux, 2))) return 0;
    if ((int64_t)4 * 16 * p->x.sC >= (1LL << 31)) return 0;
    if (p->x2.p && (!al(p->x2, 4) || (int64_t)4 * 16 * p->x2.sC >= (1LL << 31) || (p->Cin1 % 16) || p->in_scale)) return 0;
    // a tile (256 positions) must stay inside one sample when a per-(b,ci) scale is applied to the weights
    int TT = aid_pow2ceil(p->T); if (TT > 256) TT = 256;
    const int ROWS = 256 / TT;
    if (ROWS > 16) return 0;
    if (p->in_scale && (p->F % ROWS)) return 0;
    if (p->dot_ws && (p->epi != 1 || p->dot_n != aid_conv2d_dot_partials_1x1(p->B, p->Cin, p->Cout, p->F, p->T) || p->dot_n <= 0 || p->x2.p)) return 0;
    int rc;
    if (gemm && p->B * p->T <= 128 && (p->T % 4) == 0 && (p->Cin % 2) == 0 && p->ws) {
        // skinny GEMM: a wave per 32 rows and K range; enough K ranges for ~2000 waves, as the scratch allows (16 at most)
        const int nt = aid_cdiv(p->B * p->T, 32);
        const int rowtiles = p->Cout_pad / 32;
        int S, kps;
        gemm_k_partition(p->Cin, &S, &kps);
        auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
        if ((int64_t)S * p->B * p->Cout * p->T * 4 <= p->ws_bytes && al4(p->y) && (!p->res.p || al4(p->res))) {
            const dim3 grid((unsigned)aid_cdiv(rowtiles, 4), (unsigned)S);
            switch (nt) {
                case 1: hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, st, *p, (float*)p->ws, kps); break;
                case 2: hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, st, *p, (float*)p->ws, kps); break;
                case 3: hipLaunchKernelGGL(gemm_skinny_kernel<3>, grid, dim3(256), 0, st, *p, (float*)p->ws, kps); break;
                default: hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(256), 0, st, *p, (float*)p->ws, kps); break;
            }
            AID_CHECK_LAUNCH();
            aid_note_kernel("gemm_skinny_kernel+splitk");
            rc = aid_conv_splitk_reduce(p, (const float*)p->ws, S, st);
            return rc == AID_OK ? 1 : rc;
        }
    }
    if (gemm) {
        // grid-starved: split K over several workgroups per tile when the scratch is there (deterministic fixed-order reduction)
        int S, kps;
        gemm_k_partition(p->Cin, &S, &kps);
        auto al4 = [](const aid_view& v) { return (v.sB % 4) == 0 && (v.sC % 4) == 0 && (v.sF % 4) == 0 && (((uintptr_t)v.p) & 15) == 0; };
        if (S > 1 && !(p->ws && (int64_t)S * p->B * p->Cout * p->F * p->T * 4 <= p->ws_bytes && al4(p->y) && (!p->res.p || al4(p->res)))) S = 1;
        rc = launch_c11<2, 2, 4, 16, 16, 4>(p, st, S, kps);
        if (rc == 1000) return 0;
        return rc == AID_OK ? 1 : rc;
    }
    if (p->Cout_pad % 128 == 0)      rc = launch_c11<2, 2, 4, 16, 16, 4>(p, st);      // 128 x 256, 8 waves, 2 workgroups / CU
    else if (p->Cout_pad % 64 == 0)  rc = launch_c11<1, 2, 4, 16, 16, 4>(p, st);      //  64 x 256
    else if (p->Cout_pad % 96 == 0)  rc = launch_c11<3, 1, 4, 16, 16, 2>(p, st);      //  96 x 256, 4 waves
    else return 0;
    if (rc == 1000) return 0;
    return rc == AID_OK ? 1 : rc;
}
