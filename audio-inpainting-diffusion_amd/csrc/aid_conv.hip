// aid_conv2d: fused dilated dense convolution as an fp32-MFMA implicit GEMM for gfx950.
//
//   GEMM view:  M = Cout, N = (b, f, t) output positions, K = (ci, kh, kw).
//   One workgroup computes an M_BLK x N_BLK tile; its N tile is ROWS consecutive (b,f) rows times TT
//   consecutive t (TT = min(N_BLK, pow2ceil(T)), ROWS = N_BLK/TT), so the three kw taps of a row are
//   served from one LDS strip with a 1-sample halo each side and the five kh taps are five dilated
//   input rows.  K is walked in chunks of KC input channels:
//     stage  x-strips  [KC][ROWS][KH][TT+8]  global -> (scale, GELU) -> LDS   (prologue fused in staging:
//                                                     each element is transformed once per block, not per tap)
//     stage  weights   [KH*KW][KC][M_BLK]     pre-packed, cout-contiguous, float4 copies
//     MFMA   v_mfma_f32_32x32x2_f32: A = W[32 cout][2 ci], B = X[2 ci][32 t]; both operands are
//            conflict-free ds_read_b32 (lanes 0-31 consecutive floats, the two half-waves hit different ci).
//   Epilogue (registers -> global): y = alpha * (res_scale*res + acc*out_scale[b,co])   (or the dGELU form).
//
//   fp32 MFMA is exact fp32 FMA arithmetic (guide: cdna_hip_programming.md section 3), so parity with the
//   reference's F.conv2d is limited only by summation order.
#include "aid_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDev {
    aid_conv2d_params p;
    int tt_log2;      // log2(TT)
    int rows_log2;    // log2(ROWS)
    int tiles_t;      // ceil(T / TT)
    int nrows;        // B*F
    int nchunks;      // ceil(Cin / KC)
};

template <int KH, int KW, int MT, int NT, int WGM, int WGN, int KC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_mfma_kernel(const ConvDev a) {
    constexpr int NTHREADS = 64 * WGM * WGN;
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    constexpr int TAPS = KH * KW;
    constexpr int XOFF = 4 - KW / 2;   // LDS strip: [3]=left halo, [4..4+TT) core, [4+TT]=right halo

    const aid_conv2d_params& p = a.p;
    const int TT = 1 << a.tt_log2;
    const int ROWS = 1 << a.rows_log2;
    const int TTP = TT + 8;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                   // [KC][ROWS][KH][TTP]
    float* Ws = smem + KC * ROWS * KH * TTP;            // [TAPS][KC][M_BLK]
    int* rowinfo = (int*)(Ws + TAPS * KC * M_BLK);      // [ROWS][2] = (b, f) or b = -1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-n-tile LDS offsets of this lane's B element (without the ci / kh / kw terms)
    int xoff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + (lane & 31);
        const int rr = n >> a.tt_log2;
        const int tt = n & (TT - 1);
        xoff[j] = rr * KH * TTP + tt + XOFF;
    }
    const int ci_lane = lane >> 5;
    const int strip_ci = ROWS * KH * TTP;   // LDS stride between input channels

    const int tq_log2 = a.tt_log2 - 2;      // quads per strip = TT/4
    const int nquads = (KC * ROWS * KH) << tq_log2;
    const int nstrips = KC * ROWS * KH;

    for (int ch = 0; ch < a.nchunks; ++ch) {
        const int c0 = ch * KC;
        // ---- stage x strips ---------------------------------------------------------------------
        for (int q = tid; q < nquads; q += NTHREADS) {
            const int tq = q & ((1 << tq_log2) - 1);
            const int s = q >> tq_log2;
            const int kh = s % KH;
            const int s2 = s / KH;
            const int rr = s2 & (ROWS - 1);
            const int ci = s2 >> a.rows_log2;
            const int b = rowinfo[2 * rr];
            const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
            const int t = t0 + 4 * tq;
            const int c = c0 + ci;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b >= 0 && fi >= 0 && fi < p.F && c < p.Cin && t < p.T) {
                const float* src = p.x.p + (int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)fi * p.x.sF + t;
                v = *reinterpret_cast<const float4*>(src);
                if (p.in_scale) {
                    const float sc = p.in_scale[(int64_t)b * p.in_scale_ld + c];
                    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                }
                if (p.act == 1) { v.x = aid_gelu(v.x); v.y = aid_gelu(v.y); v.z = aid_gelu(v.z); v.w = aid_gelu(v.w); }
            }
            *reinterpret_cast<float4*>(Xs + s * TTP + 4 + 4 * tq) = v;
        }
        if (KW > 1) {
            for (int h = tid; h < 2 * nstrips; h += NTHREADS) {
                const int s = h >> 1;
                const int side = h & 1;
                const int kh = s % KH;
                const int s2 = s / KH;
                const int rr = s2 & (ROWS - 1);
                const int ci = s2 >> a.rows_log2;
                const int b = rowinfo[2 * rr];
                const int fi = rowinfo[2 * rr + 1] + (kh - KH / 2) * p.dilF;
                const int t = side ? (t0 + TT) : (t0 - 1);
                const int c = c0 + ci;
                float v = 0.f;
                if (b >= 0 && fi >= 0 && fi < p.F && c < p.Cin && t >= 0 && t < p.T) {
                    v = p.x.p[(int64_t)b * p.x.sB + (int64_t)c * p.x.sC + (int64_t)fi * p.x.sF + t];
                    if (p.in_scale) v *= p.in_scale[(int64_t)b * p.in_scale_ld + c];
                    if (p.act == 1) v = aid_gelu(v);
                }
                Xs[s * TTP + (side ? 4 + TT : 3)] = v;
            }
        }
        // ---- stage weights ------------------------------------------------------------------------
        {
            constexpr int MQ = M_BLK / 4;
            constexpr int WQ = TAPS * KC * MQ;
            for (int q = tid; q < WQ; q += NTHREADS) {
                const int m4 = q % MQ;
                const int ci = (q / MQ) % KC;
                const int tap = q / (MQ * KC);
                const float* src = p.wp + ((int64_t)tap * p.Cin_pad + c0 + ci) * p.Cout_pad + m0 + 4 * m4;
                *reinterpret_cast<float4*>(Ws + (tap * KC + ci) * M_BLK + 4 * m4) =
                    *reinterpret_cast<const float4*>(src);
            }
        }
        __syncthreads();
        // ---- MFMA over the chunk -------------------------------------------------------------------
#pragma unroll 1
        for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int tap = kh * KW + kw;
#pragma unroll
                for (int cp = 0; cp < KC / 2; ++cp) {
                    const int ci = 2 * cp + ci_lane;
                    float av[MT], bv[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        av[i] = Ws[(tap * KC + ci) * M_BLK + (wm * MT + i) * 32 + (lane & 31)];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bv[j] = Xs[ci * strip_ci + kh * TTP + xoff[j] + kw];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------------
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.Cout) continue;
                float v = acc[i][j][r];
                if (p.out_scale) v *= p.out_scale[(int64_t)b * p.out_scale_ld + m];
                if (p.epi == 1) {
                    const float u = p.aux.p[abase + (int64_t)m * p.aux.sC] * p.aux_scale[(int64_t)b * p.aux_scale_ld + m];
                    v *= aid_dgelu(u);
                }
                if (p.res.p) v += p.res_scale * p.res.p[rbase + (int64_t)m * p.res.sC];
                p.y.p[ybase + (int64_t)m * p.y.sC] = p.alpha * v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
static int pick_mblk(int Cout) { return Cout <= 32 ? 32 : (Cout <= 64 ? 64 : (Cout <= 96 ? 96 : 128)); }

extern "C" void aid_conv2d_pack_dims(int Cin, int Cout, int* Cin_pad, int* Cout_pad) {
    const int mb = pick_mblk(Cout);
    *Cin_pad = ((Cin + 31) / 32) * 32;
    *Cout_pad = ((Cout + mb - 1) / mb) * mb;
}

template <int KH, int KW, int MT, int NT, int WGM, int WGN, int KC>
static int launch_cfg(const aid_conv2d_params* p, hipStream_t st) {
    constexpr int M_BLK = 32 * MT * WGM;
    constexpr int N_BLK = 32 * NT * WGN;
    ConvDev a;
    a.p = *p;
    int TT = aid_pow2ceil(p->T);
    if (TT > N_BLK) TT = N_BLK;
    if (TT < 4) TT = 4;
    a.tt_log2 = aid_ilog2(TT);
    const int ROWS = N_BLK / TT;
    a.rows_log2 = aid_ilog2(ROWS);
    a.tiles_t = aid_cdiv(p->T, TT);
    a.nrows = p->B * p->F;
    a.nchunks = aid_cdiv(p->Cin, KC);
    const int rgroups = aid_cdiv(a.nrows, ROWS);
    dim3 grid((unsigned)(rgroups * a.tiles_t), (unsigned)(p->Cout_pad / M_BLK));
    const size_t lds = sizeof(float) * ((size_t)KC * ROWS * KH * (TT + 8) + (size_t)KH * KW * KC * M_BLK) +
                       sizeof(int) * 2 * ROWS;
    AID_REQUIRE(lds <= 160 * 1024, "aid_conv2d: LDS tile too large");
    auto kern = conv_mfma_kernel<KH, KW, MT, NT, WGM, WGN, KC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, st, a);
    AID_CHECK_LAUNCH();
    return AID_OK;
}

template <int KH, int KW, int KC>
static int launch_m(const aid_conv2d_params* p, hipStream_t st) {
    switch (pick_mblk(p->Cout)) {
        case 32: return launch_cfg<KH, KW, 1, 2, 1, 4, KC>(p, st);
        case 64: return launch_cfg<KH, KW, 2, 2, 1, 4, KC>(p, st);
        case 96: return launch_cfg<KH, KW, 3, 2, 1, 4, KC>(p, st);
        default: return launch_cfg<KH, KW, 2, 4, 2, 2, KC>(p, st);
    }
}

extern "C" int aid_conv2d(const aid_conv2d_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AID_REQUIRE(p && p->x.p && p->y.p && p->wp, "aid_conv2d: null pointer");
    AID_REQUIRE(p->B > 0 && p->Cin > 0 && p->Cout > 0 && p->F > 0 && p->T > 0, "aid_conv2d: empty shape");
    AID_REQUIRE((p->T % 4) == 0, "aid_conv2d: T must be a multiple of 4");
    AID_REQUIRE((p->x.sB % 4) == 0 && (p->x.sC % 4) == 0 && (p->x.sF % 4) == 0 && (((uintptr_t)p->x.p) & 15) == 0,
                "aid_conv2d: input view must be 16-byte aligned with strides % 4 == 0");
    int cip, cop;
    aid_conv2d_pack_dims(p->Cin, p->Cout, &cip, &cop);
    AID_REQUIRE(p->Cin_pad == cip && p->Cout_pad == cop, "aid_conv2d: packed weight dims mismatch (use aid_conv2d_pack_dims)");
    AID_REQUIRE(p->epi == 0 || (p->epi == 1 && p->aux.p && p->aux_scale), "aid_conv2d: epi=1 needs aux + aux_scale");
    if (p->KH == 5 && p->KW == 3) return launch_m<5, 3, 4>(p, st);
    if (p->KH == 1 && p->KW == 1) {
        if (p->Cin <= 8) return launch_m<1, 1, 8>(p, st);
        return launch_m<1, 1, 32>(p, st);
    }
    aid_set_error("aid_conv2d: unsupported kernel size (5x3 and 1x1 only)");
    return AID_E_BADARG;
}
