/*
 * aid_kernels.h -- C ABI of libaid_hip.so: the MI355X (gfx950) kernels behind the EDM inpainting hot path.
 *
 * Boundary contract (last row):
 *   - plain C: raw device pointers, sizes, strides in ELEMENTS, a hipStream_t passed as void*;
 *   - every entry point returns 0 on success or a negative AID_E_* code; nothing throws, nothing allocates
 *     (workspaces are passed in), nothing synchronises the device;
 *   - all tensors are fp32, laid out [B, C, F, T] with T contiguous (stride 1); sB/sC/sF strides let a
 *     producer write straight into a slice of a concatenation buffer (replaces the reference's torch.cat /
 *     slicing copies, networks/unet_cqt_oct_with_projattention_adaLN_2.py:769-774,814,821-822);
 *   - complex CQT coefficients cross this boundary in PLANAR form [B, 2(re,im), bins, T] (replaces
 *     view_as_real/permute/contiguous, unet...py:752-753,826-827).
 *
 * The reference has no FFI: it is pure PyTorch.  Each entry point therefore cites the torch call site(s) in
 * /root/reference it replaces.  The Python binding a maintainer would add is shown in INTEGRATION.md and
 * implemented in audio_inpainting_diffusion_amd/_lib.py (ctypes).
 */
#ifndef AID_KERNELS_H
#define AID_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AID_OK 0
#define AID_E_BADARG (-1)   /* unsupported shape / null pointer / misaligned stride */
#define AID_E_LAUNCH (-2)   /* hipLaunchKernel reported an error                    */

#define AID_ABI_VERSION 14
int aid_abi_version(void);
/* last HIP error string seen by a launcher in this process (never NULL) */
const char* aid_last_error(void);
/* name of the device kernel the most recent aid_conv2d call dispatched to (host-side bookkeeping for measurements;
   bench.py groups conv launch times per kernel family with it; never NULL) */
const char* aid_last_kernel(void);

/* ---------------------------------------------------------------------------------------------------
 * 4-D fp32 view: element (b,c,f,t) lives at p[b*sB + c*sC + f*sF + t].
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    float* p;
    int64_t sB, sC, sF;
} aid_view;

/* ---------------------------------------------------------------------------------------------------
 * aid_group_stats -- BiasFreeGroupNorm statistics + adaLN modulation folded into ONE per-(b,c) scale.
 *   replaces: BiasFreeGroupNorm.forward (unet...py:147-163) and the `x*(gamma+1)` that follows it
 *             (unet...py:465,479).
 *   computes, per sample b and group g (C/groups consecutive channels, all F*T positions):
 *       mean, std (unbiased, centred)                                  -> stats[b][g] = {mean, 1/(std+eps)}
 *       scale[b][c] = gamma[c] * (1 + mod[b*mod_ld + c]) / (std + eps)   (mod may be NULL -> 1)
 *   so that norm(x)*gamma*(1+mod) == x * scale[b][c].
 *   ws: >= B*groups*AID_STATS_SPLIT*2 doubles of scratch.
 * ------------------------------------------------------------------------------------------------- */
#define AID_STATS_SPLIT 32
typedef struct {
    aid_view x;
    int B, C, F, T, groups;
    const float* gamma;        /* [C]            */
    const float* mod;          /* [B, mod_ld] or NULL */
    int64_t mod_ld;
    float eps;
    float* scale;              /* [B, C]   out   */
    float* stats;              /* [B, groups, 2] out (mean, 1/(std+eps)) */
    double* ws;                /* scratch: [B*groups][max(ws_n, AID_STATS_SPLIT)][2] doubles */
    int ws_n;                  /* 0: the kernel reads x and reduces it itself.  > 0: `ws` already holds ws_n partial (sum, sum of squares) pairs per
                                  (sample, group), written by the epilogue of the aid_conv2d that produced x (its stat_ws / stat_n): the read pass is skipped */
} aid_group_stats_params;
int aid_group_stats(const aid_group_stats_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_conv2d -- fused dilated dense convolution as fp32-MFMA implicit GEMM (the 99 %-of-FLOPs kernel).
 *   replaces: Conv2d.forward / F.conv2d(padding="same", dilation=(d,1)) (unet...py:79-88) together with the
 *             elementwise work around it in ResnetBlock.forward (unet...py:472-482, 456, 488-491),
 *             the pyramid projection + combine (unet...py:794), the 1x1 channel projections, and (with
 *             F=1, KH=KW=1) the qk Conv1d of TimeAttentionBlock (unet...py:355).
 *       xin  = act( x[b,ci,f',t'] * in_scale[b,ci] )            act: 0 none, 1 erf-GELU; in_scale may be NULL
 *       acc  = sum_{ci,kh,kw} w[co,ci,kh,kw] * xin[b,ci,f+(kh-KH/2)*dilF, t+(kw-KW/2)]   (zero 'same' padding)
 *       y    = alpha * ( res_scale * res[b,co,f,t] + acc * out_scale[b*out_scale_ld + co] )
 *              (res may be NULL -> 0 ; out_scale may be NULL -> 1)
 *   epilogue extension for the input-VJP (guidance branch), selected by `epi`:
 *       epi = 0 : as above
 *       epi = 1 : y = alpha * acc * out_scale * gelu'(aux[b,co,f,t] * aux_scale[b,co])   (dGELU epilogue)
 *   Winograd path (5x3, no in-kernel prologue, Cin % 4 == 0, Cout >= 64, T % 4 == 0): when `wp_wino` is given the
 *   launcher uses F(4,3) along T -- 6 MFMAs per 4 output samples (2x fewer than the direct form).  U = G w with
 *   G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]] packed as 30 "taps"
 *   xi*KH+kh; V = B^T d either formed from the LDS strip at fragment-load time or supplied by the caller (x_wino);
 *   output transform in the epilogue.  Same exact-fp32 MFMA; results differ from the direct form only by fp32
 *   rounding (tests: <= 2e-6 rel-L2, about 2x the error of direct accumulation).
 *   weights are PRE-PACKED by the host: wp[KH*KW][Cin_pad][Cout_pad], cout contiguous, zero padded
 *   (Cin_pad % 32 == 0, Cout_pad a multiple of the M tile the launcher picks for Cout) -- see aid_conv2d_pack_dims.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view x, y, res, aux;
    const float* wp;
    const float* in_scale;   int64_t in_scale_ld;
    const float* out_scale;  int64_t out_scale_ld;
    const float* aux_scale;  int64_t aux_scale_ld;
    int B, Cin, Cout, F, T;
    int Cin_pad, Cout_pad;
    int KH, KW, dilF;
    int act, epi;
    float alpha, res_scale;
    const float* wp_wino;     /* optional: Winograd F(4,3) pack of the same 5x3 weights, 30 taps (see above); with x_wino = 2: the F(8,3) pack, 50 taps xi*5+kh */
    int wino_taps;            /* 30, or 50 with x_wino = 2 (0 when wp_wino is NULL) */
    int x_wino;               /* 1: `x` is the F(4,3) INPUT TRANSFORM of the activations, written by aid_scale_act(wino=1):
                                 [B, Cin, F, 6, T/4] (sF = 6*T/4 ...), V = B^T d per group of 4 samples; needs wp_wino with 30 taps
                                 and aid_conv2d_wino_input_supported(...) != 0.  The kernel then stages and multiplies only.
                                 2: `x` is the F(8,3) input transform [B, Cin, F, 10, T/8] (aid_scale_act wino = 2): 10 MFMAs per 8 outputs
                                 (0.417x the direct form); needs the 50-tap pack and aid_conv2d_wino_form(...) == 8.  fp32 error about 2x that of
                                 F(4,3) (3e-6 rel-L2 per layer at Cin = 128; profiles/r04_wino_fm3_error.txt).
                                 3: the NON-FUSED 2-D form F(4,5) x F(4,3) (3.0 products per output; csrc/aid_wino2d.hip): x.p is V [48][Cin][N] written by
                                 aid_scale_act(wino = 3) (x's strides are not used), wp_wino the 48-plane pack U = GF w GT^T (aid_pack_conv_weight wpw2),
                                 wino_taps = 48, and `ws` MUST hold 48 * Cout * N floats for M = U V: aid_wino2d_gemm, then the output-transform pass with the
                                 epilogue below; needs aid_conv2d_wino2d_supported(...) != 0 (Cout % 128 == 0, Cin % 16 == 0, F % dilF == 0, T % 16 == 0).
                                 fp32 error 2-4e-6 rel-L2 per layer (profiles/r04_wino2d_fm5_error.txt).
                                 4: the same three passes with F(8,3) along T -- F(4,5) x F(8,3), 80 planes, 2.5 products per output, V / M 2.5 x the activation:
                                 x.p is V [80][Cin][N/2] written by aid_scale_act(wino = 4), wp_wino the 80-plane pack (aid_pack_conv_weight wpw3), wino_taps = 80,
                                 `ws` holds 80 * Cout * (N/2) floats (N = aid_conv2d_wino2d_positions: groups of 8 samples halve it); additionally T % 32 == 0.
                                 fp32 error 1.0e-5 (Cin = 128) / 1.3e-5 (Cin = 256) rel-L2 per layer (profiles/r06_wino2d_f45x83_error.txt). */
    float* ws; int64_t ws_bytes; /* optional scratch (the library never allocates): lets grid-starved 1x1 GEMMs (the qk projections:
                                 B*T columns only, K of several thousand) split K over up to 8 workgroups per tile; partial sums
                                 go to ws[S][B][Cout][F][T] and a second kernel reduces them in a FIXED order (deterministic) and
                                 applies the epilogue.  NULL / too small: single-pass kernel.
                                 5x3 layers on the row-shared F(4,3) kernel at B = 1 (aid_conv2d_wino_split_ws_bytes(...) > 0): the K axis of
                                 every tile is shared by two workgroups; ws then holds [AID_CONV2D_SPLIT_FLAG_BYTES of flags][partial accumulators], and its
                                 first AID_CONV2D_SPLIT_FLAG_BYTES bytes must be ZERO before the first use (the kernel leaves them zero; the LAST word of that region is a sticky
                                 error word: a workgroup whose bounded wait for its partner ever times out sets it, and from then on every split launch on this
                                 scratch writes NaN tiles instead of trusting flags that may be stale -- re-zero the scratch to recover);
                                 one ws per stream.  NULL / too small: no split. */
    double* dot_ws; int dot_n;   /* optional (input-VJP, epi = 1 on the F(4,3) path or on a 1x1 layer, see aid_conv2d_dot_partials_1x1): the epilogue also reduces <y, aux> per
                                 (sample, channel group of Cout/8) over its tile and writes one partial per tile,
                                 dot_ws[(b*8 + g) * dot_n + tile_in_sample] -- the aid_group_dot pass over the dgrad output
                                 folded into the conv that produces it.  dot_n must equal aid_conv2d_dot_partials(...). */
    double* stat_ws; int stat_n; /* optional (forward, epi = 0, Winograd-domain input on the row-shared kernel only; not together with dot_ws): the epilogue
                                 also reduces (sum y, sum y^2) per (sample, channel group of Cout/8) over its tile:
                                 stat_ws[((b*8 + g) * stat_n + tile) * 2 + {0,1}] -- the read pass of the NEXT layer's aid_group_stats folded into
                                 the conv that produces its input (aid_group_stats ws_n = stat_n).  stat_n must equal aid_conv2d_stat_partials(...). */
    aid_view x2; int Cin1;       /* optional (1x1 only, no prologue): input channels [Cin1, Cin) are read from x2 (channel c of x2 = input channel
                                 Cin1 + c); Cin1 and Cin - Cin1 multiples of 16.  One GEMM over a K axis that lives in two tensors: the input
                                 gradients of a ResnetBlock's proj_in and res_conv (unet...py:414-415, :488-491), which both flow into dL/d(block
                                 input), as ONE launch on the stacked transposed weights instead of two read-modify-write passes over it. */
    int fin_mode;                /* optional, row-shared Winograd kernels and the output pass of the 2-D form (aid_conv2d_fin_supported(...) != 0): the LAST tile / block of a sample to finish also folds
                                 the sample's epilogue partials, in the fixed order of the kernels it replaces (bit-identical results) --
                                 1 (with stat_ws): what aid_group_stats(ws_n = stat_n) does: fin_scale[b, c] = fin_gamma[c] (1 + fin_mod[b, c]) / (std_g + fin_eps),
                                    fin_stats[b, g] = (mean, 1 / (std + eps)) when given -- the NEXT layer's aid_group_stats call is then not made at all;
                                 2 (with dot_ws): what the first kernel of aid_norm_bwd(ws_n = dot_n) does: fin_scale[b*8 + g] = <gd, x>_g inv / ((n - 1) std) from the
                                    forward statistics fin_stats (input) -- aid_norm_bwd is then called with coef_ready = 1 and fin_scale must be its coefficient
                                    scratch (the B*8 floats behind the partials in its `ws`).
                                 0: none. */
    unsigned* fin_count;         /* [B] arrival counters, ZERO before the first use (the kernel leaves them zero); one array per stream */
    const float* fin_gamma;      /* mode 1: [Cout] */
    const float* fin_mod; int64_t fin_mod_ld;   /* mode 1: optional [B, fin_mod_ld] */
    float fin_eps;
    float* fin_scale;
    float* fin_stats;
} aid_conv2d_params;
int aid_conv2d(const aid_conv2d_params* p, void* stream);
/* partials per (sample, group) when dot_ws is asked of a 1x1 layer (epi = 1, no residual; direct-to-LDS kernel); 0: not available */
int aid_conv2d_dot_partials_1x1(int B, int Cin, int Cout, int F, int T);
/* 1 when aid_conv2d accepts the x2 / Cin1 option (K axis in two tensors) for this 1x1 shape */
int aid_conv2d_x2_supported(int Cin, int Cin1, int Cout, int F, int T);
/* non-zero when a 5x3 layer of this shape can take Winograd-domain input (x_wino = 1) */
int aid_conv2d_wino_input_supported(int Cin, int Cout, int T);
/* the same question with the layer's geometry: T = 16 layers are served only when the row-shared tiles fit F and the dilation */
int aid_conv2d_wino_input_ok(int B, int Cin, int Cout, int F, int T, int dilF);
/* which Winograd-domain input a 5x3 layer of this launch shape should be given: 8 -> x_wino = 2 (F(8,3)), 4 -> x_wino = 1 (F(4,3)), 0 -> plain
   activations.  F(8,3) takes the launch when its row-shared tiles fit (T % 32 == 0, F % dilF == 0, <= 15 % padding rows) and its twice-as-large
   tiles still quantise better over the CUs than F(4,3)'s (a function of the launch shape, B included; see wino_form_choice in csrc/aid_conv_wino.hip). */
int aid_conv2d_wino_form(int B, int Cin, int Cout, int F, int T, int dilF);
/* 1 when a 5x3 layer of this per-sample shape CAN take x_wino = 2 (the F(8,3) row-shared tiles fit); aid_conv2d_wino_form() then says whether it should */
int aid_conv2d_wino8_supported(int Cin, int Cout, int F, int T, int dilF);
/* bytes of `ws` a 5x3 x_wino layer of this shape wants for its split-K instance (B = 1 launches with few tiles); 0: the shape is not split */
#define AID_CONV2D_SPLIT_FLAG_BYTES 4096
int64_t aid_conv2d_wino_split_ws_bytes(int B, int Cin, int Cout, int F, int T, int dilF);
/* 1 when the kernel that takes a 5x3 layer of this launch shape with this x_wino (1 / 2 / 3) honours fin_mode (the row-shared kernels; the 2-D form's output pass) */
int aid_conv2d_fin_supported(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino);
/* number of per-tile partial dots per (sample, group) the F(4,3) / F(8,3) epilogue writes for this shape (x_wino as in aid_conv2d_params); 0 = not supported */
int aid_conv2d_dot_partials(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino);
/* number of per-tile (sum, sum of squares) partials per (sample, group) for stat_ws; 0 = the kernel that takes this shape does not write them */
int aid_conv2d_stat_partials(int B, int Cin, int Cout, int F, int T, int dilF, int x_wino);
/* x_wino = 3 (2-D Winograd form): 1 when the layer shape is supported; positions per transform plane N for a launch of B samples (0: unsupported) */
int aid_conv2d_wino2d_supported(int Cin, int Cout, int F, int T, int dilF);
int64_t aid_conv2d_wino2d_positions(int B, int F, int T, int dilF);
/* 1 when a launch of this shape SHOULD take the 2-D form (measured per layer against the fused 1-D kernels; a function of the launch shape) */
int aid_conv2d_wino2d_wanted(int B, int Cin, int Cout, int F, int T, int dilF);
/* the T form the 2-D form of such a launch should take: 0 = not a 2-D launch (aid_conv2d_wino2d_wanted == 0), 4 = F(4,5) x F(4,3) (x_wino = 3), 8 = F(4,5) x F(8,3)
   (x_wino = 4; T % 32 == 0).  A function of the launch shape.  (ABI 14) */
int aid_conv2d_wino2d_tform(int B, int Cin, int Cout, int F, int T, int dilF);
/* padded dims the packed weight buffer must have for a given (Cin, Cout) */
void aid_conv2d_pack_dims(int Cin, int Cout, int* Cin_pad, int* Cout_pad);

/* ---------------------------------------------------------------------------------------------------
 * aid_wino2d_gemm -- the batched fp32-MFMA GEMM at the centre of the 2-D Winograd form F(4,5) x F(4,3) of the dilated 5x3 convolution
 *   (aid_conv2d x_wino = 3 runs it followed by the output-transform pass; exposed on its own for tests and measurements).
 *   replaces: the multiply-accumulate work of Conv2d.forward / F.conv2d(dilation=(d,1)) (unet...py:79-88, :433-436, :472-482) on the C >= 128 layers.
 *   For every transform index xi < nxi (48; 80 for the F(4,5) x F(8,3) form):   M[xi][co][n] = sum_ci U[xi][ci][co] * V[xi][ci][n]
 *   U [nxi][Cin_pad][Cout_pad] (aid_pack_conv_weight wpw2 / wpw2T), V [nxi][Cin][N] (aid_scale_act wino = 3), M [nxi][Cout][N]; N % 4 == 0,
 *   Cin a multiple of 16, Cout_pad a multiple of 128, every plane below 4 GiB.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* U; const float* V; float* M;
    int nxi, Cin, Cout, Cin_pad, Cout_pad;
    int64_t N;
    int variant;               /* 0: the product instance; > 0: tile-shape experiments (tools/w2d_probe.py) */
} aid_wino2d_gemm_params;
int aid_wino2d_gemm(const aid_wino2d_gemm_params* p, void* stream);
/* The two launches of aid_conv2d(x_wino = 3) as separate calls on the SAME parameter block (a launch plan that wants the MFMA-bound GEMM and the
 * HBM-bound output pass as separate nodes: per-kernel timing, different streams): _gemm = M = U V into p->ws (aid_wino2d_gemm),
 * _output = the output transform y = AF^T M AT with the epilogue (gate, residual, dGELU, statistics / dot partials) from p->ws.
 * _gemm followed by _output on one stream is exactly aid_conv2d(p).   replaces: the same F.conv2d call sites as aid_conv2d (unet...py:433-436, :472-482). */
int aid_conv2d_wino2d_gemm(const aid_conv2d_params* p, void* stream);
int aid_conv2d_wino2d_output(const aid_conv2d_params* p, void* stream);
/* LABELLED VARIANT, off by default (never the measured headline): pieces = 6 makes the GEMM of the 2-D form split both fp32 operands into three bf16
 * pieces (exactly) and issue the six largest cross products on the bf16 matrix pipe with fp32 accumulation (aid_wino2d_gemm variants 100 / 101);
 * pieces = 0 restores the fp32-MFMA kernel.  Process-wide.  Error study: profiles/r05_bf16split_error.txt (1.8e-6 per layer at K = 256; fp32 MFMA 4.3e-6). */
int aid_wino2d_set_split(int pieces);

/* ---------------------------------------------------------------------------------------------------
 * aid_resample -- 8-tap cubic FIR 2:1 resampling along T with reflect padding.
 *   replaces: UpDownResample.forward (unet...py:549-580; taps :514-515).  The reference runs this as a
 *             dense conv with a diagonal [F,F,8] weight; here it is the 16-FLOP/output FIR.
 *   up = 0: y[..., j] = sum_k h[k] * xr[2j + k - 3]                    T_out = T/2
 *   up = 1: transposed conv, stride 2, crop 7 (no x2 gain)              T_out = 2T
 *   adjoint = 1 computes the exact transpose of the selected map (used by the input-VJP).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view x, y;
    int B, C, F, T;        /* T = length of the tensor passed as x (for adjoint=1: the incoming gradient) */
    int up, adjoint;
    int accumulate;        /* 1: y += result (gradient accumulation), 0: y = result */
} aid_resample_params;
int aid_resample(const aid_resample_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_time_attention -- softmax(q k^T * F^-1/2) v over the time axis, per (sample, head).
 *   replaces: the einsum/softmax/einsum core of TimeAttentionBlock.forward (unet...py:358-374).
 *   qk : [B, H*2F, T]  (rows h*2F .. h*2F+F-1 = q of head h, next F rows = k)   -- output of the qk GEMM
 *   v  : [B, H, F, T]  (the projected input itself, unet...py:353)
 *   out: [B, H, F, T]
 *   T <= 128.  If `probs` is non-NULL the softmax matrix [B,H,T,T] is also written (saved for the VJP).
 *   QK^T, PV and the VJP products run on v_mfma_f32_32x32x2_f32; the softmax reduces over registers, one wavefront shuffle and LDS.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* qk; const float* v; float* out; float* probs;
    int B, H, F, T;
    float scale;
    const float* bias;        /* optional [H, T, T] additive logit bias, applied BEFORE the scale: softmax((q k^T + bias) * scale)
                                 -- the relative-position bias of attention_dict.use_rel_pos (unet...py:266-312,364); NULL: none */
} aid_attention_params;
int aid_time_attention(const aid_attention_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_embed -- RFF noise-level embedding + 3-layer ReLU MLP.
 *   replaces: RFF_MLP_Block.forward (unet...py:184-211).   sigma[B] -> emb[B, emb_dim]
 * aid_modulation -- every affine/gate Linear of the network as ONE batched product.
 *   replaces: the ~194 `affine(sigma)` / `gate(sigma)` Linear calls per evaluation (unet...py:461-462,476-477).
 *   mod[b, j] = sum_e emb[b,e] * W[j,e] + bias[j],  W = all affine/gate weights stacked row-wise [N, E].
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* sigma;                /* [B] (the c_noise value)            */
    const float* rff_freq;             /* [rff]                              */
    const float* w0; const float* b0;  /* [h0, 2*rff], [h0]                   */
    const float* w1; const float* b1;  /* [h1, h0]                            */
    const float* w2; const float* b2;  /* [E, h1]                             */
    float* emb;                        /* [B, E]                              */
    int B, rff, h0, h1, E;
} aid_embed_params;
int aid_embed(const aid_embed_params* p, void* stream);

typedef struct {
    const float* emb; const float* W; const float* bias; float* mod;
    int B, E, N;
} aid_modulation_params;
int aid_modulation(const aid_modulation_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * CQT (octave-mode NSGT).  replaces the external cqt_nsgt_pytorch calls CQT_nsgt.fwd / .bwd /
 * .apply_hpf_DC (call sites unet...py:743,841; testing/edm_sampler_inpainting.py:63,123).
 * The length-L real FFTs on either side are aid_fft_pass (mixed-radix Stockham, one launch per radix; below).
 *
 * Band table (device, built once by the host from audio_inpainting_diffusion_amd/cqt.py):
 *   band k of octave o:  centre bin rc[k], window length Lg[k], window samples g[goff[k] .. goff[k]+Lg[k])
 *   (analysis window) or the dual window times M_k (synthesis), sampled at offsets j = -Lg/2 .. Lg-Lg/2-1.
 *
 * aid_cqt_analysis : spec[B, Lh] complex (interleaved re,im; Lh = L/2+1)  ->  coef octave tensors.
 *     c_k[t] = IFFT_T( wrap_T( spec[rc_k + j] * g_k[j] ) ) * in_scale[b]
 *     written to out[b, 0/1, bin, t] of the octave's planar view.
 * aid_cqt_synthesis: coef octave views -> band spectra ws[b][k][T_o] = FFT_T(c_k)[.] (complex interleaved)
 * aid_cqt_gather   : Y[b][v] = hpf[v] * ( cskip[b] * X[b][v] + cout[b] * sum_k ws[b][k][(v-rc_k) mod T_o] * gdM_k[v-rc_k] )
 *     (X, cskip, cout, hpf optional) -- deterministic overlap-add, no atomics.
 *
 * The exact adjoints needed by the input-VJP re-use the same three kernels with other window tables:
 *   adjoint(synthesis+gather) = aid_cqt_analysis with g := gdM and unnormalized = 1;
 *   adjoint(analysis)         = aid_cqt_synthesis + aid_cqt_gather with gdM := g / T_k.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int n_oct, bins;                 /* octaves, bins per octave                       */
    const int* rc;                   /* [n_oct*bins] centre bins                        */
    const int* Lg;                   /* [n_oct*bins] window lengths                     */
    const int* goff;                 /* [n_oct*bins] offsets into g                     */
    const float* g;                  /* concatenated windows                            */
    const int* T_oct;                /* [n_oct] host-side copy is passed in T_host      */
    const float* twiddle;            /* [Tmax/2] complex interleaved exp(-2 pi i m/Tmax) */
    int Tmax;
} aid_cqt_tables;

#define AID_CQT_MAX_OCT 12
typedef struct {
    aid_cqt_tables tab;
    int T_host[AID_CQT_MAX_OCT];     /* octave lengths (host copy)                      */
    aid_view oct[AID_CQT_MAX_OCT];   /* per-octave planar views [B,2,bins,T_o] (sC = stride between re/im planes) */
    const float* spec;               /* analysis: in  [B, Lh, 2]                         */
    float* band_ws;                  /* synthesis: out [B, sum_o bins*T_o, 2]            */
    const float* in_scale;           /* analysis: per-sample scale [B] or NULL          */
    int B, Lh;
    int unnormalized;                /* analysis: 1 = skip the 1/T of the inverse FFT (adjoint of the synthesis FFT) */
} aid_cqt_params;
int aid_cqt_analysis(const aid_cqt_params* p, void* stream);
int aid_cqt_synthesis(const aid_cqt_params* p, void* stream);

typedef struct {
    const float* band_ws;            /* [B, sum_o bins*T_o, 2]                           */
    const int* kfirst; const int* kcount;   /* [Lh] first covering band and count        */
    const int* rc; const int* Lg; const int* goff; const int* woff; const int* Tk;  /* per band; woff = offset of band in band_ws (complex elems) */
    const float* gdM;                /* concatenated dual windows times M_k              */
    const float* X;                  /* [B, Lh, 2] or NULL                               */
    const float* cskip; const float* cout;  /* [B] or NULL (-> 0 / 1)                    */
    const float* hpf;                /* [Lh] or NULL  (any real per-bin multiplier)      */
    float* Y;                        /* [B, Lh, 2]                                       */
    int B, Lh; int64_t ws_per_b;
    const float* band_scale;         /* [Lh] or NULL: extra per-bin multiplier of the band sum only (adjoint paths) */
} aid_cqt_gather_params;
int aid_cqt_gather(const aid_cqt_gather_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_fft_pass -- one radix-R pass of a mixed-radix Stockham FFT of length N = prod(R_i) (R <= 32), batched.
 *   replaces: the two length-L real FFTs per evaluation that cqt_nsgt_pytorch delegates to torch.fft
 *             (L = 184184 = 8*7*11*13*23 for the shipped configurations; any smooth N with prime factors <= 31).
 *   A full transform = one launch per radix, ping-ponging between two buffers (host drives the passes):
 *     pass with radix R, Ns = product of the radices already applied:
 *        v[r] = in[j + r*N/R] * W_N^{ r*k*N/(Ns*R) },  k = j mod Ns          (j < N/R)
 *        out[(j/Ns)*Ns*R + k + q*Ns] = sum_r v[r] * W_R^{q r}
 *   in_mode : 0 complex [B,N]; 1 real [B,N] (imag = 0);  2 half spectrum [B,N/2+1] extended Hermitian-ly
 *   out_mode: 0 complex [B,N]; 1 real part only [B,N] (times out_scale); 2 first N/2+1 bins [B,N/2+1]
 *   sign = -1 forward, +1 inverse (unnormalised; fold 1/N into out_scale).
 *   twiddle: [N] complex exp(-2 pi i m / N) (fp64-computed on the host).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* in; float* out; const float* twiddle;
    int B, N, R, Ns;
    int in_mode, out_mode;
    float sign, out_scale;
} aid_fft_pass_params;
int aid_fft_pass(const aid_fft_pass_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Sampler element-wise kernels (replace the tensor expressions of Sampler.predict / get_score,
 * testing/edm_sampler_inpainting.py:141-147, 204-251, 343).  All arrays [B, L]; per-sample scalars [B].
 *
 * aid_axpby   : out = a[b]*x + b_[b]*y                       (churn add :214; generic)
 * aid_score_step : data-consistency projection + ODE direction + Euler proposal in one pass
 *       xh  = smask*yobs + (1-smask)*xhat                     (:343)      (smask NULL -> xh = xhat)
 *       d   = (x - xh) / t[b]                                 (:105,:230  d = -t*score)
 *       mode 0:  xnext = x + h[b]*d ;  dout = d               (Euler proposal :240 / final step :251)
 *       mode 1:  xnext = x0 + h[b]*0.5*(d0 + d)               (Heun combine :247; x0,d0 from the first eval)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; const float* y; float* out;
    const float* a; const float* b;   /* per-sample [B]; NULL: the host scalar a_host / b_host for every sample */
    int B; int64_t L;
    float a_host, b_host;
} aid_axpby_params;
int aid_axpby(const aid_axpby_params* p, void* stream);

typedef struct {
    const float* x; const float* xhat; const float* yobs; const float* smask; int64_t smask_sB;
    const float* x0; const float* d0;
    const float* t; const float* h;   /* per-sample [B]; NULL: the host scalars t_host / h_host (one schedule for the batch) */
    float* xnext; float* dout; float* xh_out;
    int B; int64_t L; int mode;
    float t_host, h_host;
} aid_score_step_params;
int aid_score_step(const aid_score_step_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Input-VJP helpers (reconstruction-guidance branch: torch.autograd.grad(norm, x) at
 * testing/edm_sampler_inpainting.py:78-81 walks back through the whole denoiser).
 *
 * aid_group_dot  : partial sums of <u, v> per (sample, channel group) into ws (same layout as aid_group_stats).
 * aid_norm_bwd   : backward of the group-std normalisation feeding a conv (x -> x*scale[b,c], scale ~ 1/(std+eps)):
 *       out (+)= gd - coef[b,g] * (x - mean[b,g]) + a * gy
 *       coef = <gd, x>_group * inv / ((n-1) * std),  inv = 1/(std+eps) from aid_group_stats' `stats`,
 *       gd = dgrad conv output (already times scale and GELU'), gy optional skip-path gradient.
 * aid_time_attention_bwd : gradients of aid_time_attention w.r.t. qk and v given d(out) and the saved probs.
 * aid_guidance_seed : g = d/dxhat || y - mask*xhat ||_2 = -mask * (y - mask*xhat) / norm[b]   (per item; :65-75)
 * aid_row_norm   : out[b] = || x[b,:] ||_2                                                   (:83)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view u, v;
    int B, C, F, T, groups;
    double* ws;
} aid_group_dot_params;
int aid_group_dot(const aid_group_dot_params* p, void* stream);

typedef struct {
    aid_view gd, x, gy, out;      /* gy.p may be NULL */
    int B, C, F, T, groups;
    const float* stats;           /* [B, groups, 2] (mean, 1/(std+eps)) from the forward aid_group_stats */
    const double* ws;             /* partial dots from aid_group_dot (ws_n = 0) or from the conv epilogue (ws_n partials per (b,g));
                                     B*groups floats of scratch follow the partials */
    float eps, a;
    int accumulate;
    int ws_n;
    aid_view wout;                /* optional second output (T % 16 == 0): the F(4,3) input transform [B, C, F, 6, T/4] of out * wscale[b,c] -- what
                                     aid_scale_act(wino=1) would write for the dgrad conv of the layer below (its gate pre-pass folded into this pass) */
    const float* wscale; int64_t wscale_ld;   /* [B, wscale_ld] or NULL (-> 1) */
    int wform;                    /* Winograd form of wout: 0 / 1 = F(4,3) as above; 2 = F(8,3), wout rows [10][T/8] (aid_scale_act wino = 2);
                                     3 / 4 (ABI 14): the 2-D forms -- wout.p is V [48 | 80][C][N] (wout's strides unused), exactly what aid_scale_act(wino = 3 | 4,
                                     act = 0, scale = wscale, dilF = wdil) would write from `out`: this pass then IS the input pass of the dgrad conv of the layer
                                     below (reads gd, x, gy once -- 1.125 x with the halo rows -- writes out and V); accumulate must be 0 */
    int coef_ready;               /* 1: the B*groups coefficients behind the partials in `ws` were already written by the conv that produced the partials
                                     (aid_conv2d fin_mode = 2): the coefficient kernel is not launched */
    int wdil;                     /* wform = 3 | 4 only: the dilation of the 5x3 layer that will read V */
} aid_norm_bwd_params;
int aid_norm_bwd(const aid_norm_bwd_params* p, void* stream);

typedef struct {
    const float* qk; const float* v; const float* probs; const float* gout;
    float* gqk; float* gv;        /* gqk written; gv accumulated (+=) when accumulate_gv != 0 */
    int B, H, F, T;
    float scale;
    int accumulate_gv;
    float* ws;                    /* scratch [B,H,T,T] (dS) */
} aid_attention_bwd_params;
int aid_time_attention_bwd(const aid_attention_bwd_params* p, void* stream);

typedef struct {
    const float* xhat; const float* y; const float* mask; int64_t mask_sB;   /* mask NULL: all ones (operator degradations) */
    float* g; float* norm;        /* g [B,L], norm [B] */
    int B; int64_t L;
    int norm_type;                /* tester.posterior_sampling.norm (:72-75): 2 = L2, 1 = L1, 3 = "smoothl1" (summed over the item) */
    float beta;                   /* smoothl1_beta (norm_type 3) */
} aid_guidance_seed_params;
int aid_guidance_seed(const aid_guidance_seed_params* p, void* stream);

/* aid_guidance_step : the guidance update of one evaluation in one launch (:83-97), per item:
 *       normguide = ||g[b]||_2 * inv_sqrt_len ;  s = coef / (normguide + eps) ;  out = xhat - s*g
 *   coef = t_i * xi (host), inv_sqrt_len = 1/sqrt(audio_len), eps = 1e-6; optional s_out [B] and step_out = s*g [B,L] (rid buffers). */
typedef struct {
    const float* xhat; const float* g; float* out; float* step_out; float* s_out;
    int B; int64_t L;
    float coef, inv_sqrt_len, eps;
} aid_guidance_step_params;
int aid_guidance_step(const aid_guidance_step_params* p, void* stream);

/* aid_set_rows : out[i*ld + b] = v[i], i < n <= 8, b < B -- the per-evaluation EDM scalars (c_noise, c_in, c_skip, c_out: edm.py:97-128,
 *   computed on the host in the reference's float32 arithmetic) broadcast to the [B] vectors the fused kernels read, in one launch. */
typedef struct { float* out; int64_t ld; int B, n; float v[8]; } aid_set_rows_params;
int aid_set_rows(const aid_set_rows_params* p, void* stream);

typedef struct { const float* x; float* out; int B; int64_t L; } aid_row_norm_params;
int aid_row_norm(const aid_row_norm_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_resample_poly -- rational-ratio (orig -> new) polyphase sinc resampling of whole waveforms.
 *   replaces: torchaudio.functional.resample as called by resample_batch (utils/training_utils.py:140-212; used by
 *             Tester.resample_audio, testing/tester_inpainting.py:558-560): 44.1k -> 22.05k (2:1), 48k -> 22.05k (320:147),
 *             48k -> 44.1k (160:147), anything -> fs_target.
 *   y[b, i*new + j] = sum_{k < K} kernel[j*K + k] * xpad[b, i*orig + k],  xpad = x zero-padded by `width` on the left,
 *   K = 2*width + orig; orig/new already divided by their gcd; Lout <= ceil(new*L/orig) outputs are written.
 *   The kernel table (Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99) is the host's (harness.py).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; float* y; const float* kernel;
    int64_t x_ld, y_ld;           /* row strides of x [B, L] and y [B, Lout] */
    int64_t L, Lout;
    int B, orig_freq, new_freq, width, K;
} aid_resample_poly_params;
int aid_resample_poly(const aid_resample_poly_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_stft_frames / aid_stft_ola -- the STFT-domain masking operator of spectrogram inpainting and its adjoint.
 *   replaces: Sampler.apply_spectral_mask (testing/edm_sampler_inpainting.py:271-290):
 *       x -> pad to a multiple of n_fft -> torch.stft(n_fft, hop, win, hann, center=True, reflect) -> * mask[F,T]
 *         -> torch.istft -> crop to L,
 *     used as the degradation inside the guidance norm (:65-75, through torch.autograd) and in the projection
 *     y + x - A(x) (:360).  adjoint = 1 gives A^T (what autograd would compute), reusing both kernels.
 *   aid_stft_frames: x [B,L] -> frames [B, n_frames, n_fft]   (window, FFT, mask, inverse FFT, window / n_fft)
 *   aid_stft_ola   : frames -> out [B,L] = c0 * (overlap-add [* 1/envelope when adjoint = 0]) + add1 + add2
 *   Lp = L + (n_fft - L % n_fft) (the reference always pads, :283), n_frames = 1 + Lp/hop,
 *   window [n_fft] (win_length centre-padded to n_fft by the host), twiddle [n_fft/2] complex exp(-2 pi i m/n_fft),
 *   mask [F = n_fft/2+1 rows, leading dimension mask_ld >= n_frames], per-sample stride mask_sB (0 = shared),
 *   inv_env [Lp] = 1 / sum_n window^2 (the istft normalisation, trimmed by n_fft/2).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; float* frames; float* out;
    const float* window; const float* twiddle; const float* inv_env;
    const float* mask; int64_t mask_sB; int64_t mask_ld;
    const float* add1; const float* add2; float c0;
    int B; int64_t L; int64_t Lp;
    int n_fft, hop, n_frames, adjoint;
} aid_stft_params;
int aid_stft_frames(const aid_stft_params* p, void* stream);
int aid_stft_ola(const aid_stft_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_scale_act -- h = act(x * scale[b,c]) on [B,C,F,T] views (act: 0 none, 1 erf-GELU).
 *   The normalise -> modulate -> GELU prologue of a dilated step (unet...py:475-482) evaluated ONCE per element
 *   into a scratch tensor; the 5x3 conv then stages plain copies (inside the conv's LDS staging the same GELU
 *   would be re-evaluated for each of the 5 dilated rows and each Cout tile).
 *   wino = 1 additionally applies the Winograd F(4,3) input transform in the same pass (the conv kernel then neither
 *   transforms at fragment-load time nor needs halo samples):
 *     V0 = 4d0-5d2+d4, V1 = (d3+d4)-4(d1+d2), V2 = (d4-d3)+4(d1-d2), V3 = (d4-d2)+2(d3-d1), V4 = (d4-d2)-2(d3-d1), V5 = 4d1-5d3+d5
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view x, y;
    const float* scale; int64_t scale_ld;
    int B, C, F, T, act;
    int wino;                 /* 1: write the F(4,3) input transform of h along T instead of h itself: y is [B,C,F,6,T/4],
                                 y[..,xi,g] = (B^T d)[xi], d = h[4g-1 .. 4g+4] (zero outside the row); T % 16 == 0.
                                 2: the F(8,3) input transform: y is [B,C,F,10,T/8], d = h[8g-1 .. 8g+8]; interpolation points
                                 {0, +-0.4, +-0.8, +-1.25, +-2.5, inf}, matrices in csrc/aid_wino8.h (tools/gen_wino8.py)
                                 3: the 2-D input transform F(4,5) x F(4,3) for aid_conv2d x_wino = 3 (matrices: csrc/aid_wino45.h, tools/gen_wino45.py):
                                 y.p is V [48][C][N] (y's strides are not used), N = aid_conv2d_wino2d_positions(B, F, T, dilF),
                                 V[xf*6 + xt][c][b*NB + (j*dilF + r)*(T/4) + g] = (BF^T h BT)[xf][xt] of the 8 x 6 patch of rows r + dilF*(4j-2 .. 4j+5),
                                 samples 4g-1 .. 4g+4 (zero outside the tensor); F % dilF == 0, T % 16 == 0
                                 4: the 2-D input transform F(4,5) x F(8,3) for aid_conv2d x_wino = 4: y.p is V [80][C][N/2],
                                 V[xf*10 + xt][c][b*NB + (j*dilF + r)*(T/8) + g] of the 8 x 10 patch of rows r + dilF*(4j-2 .. 4j+5), samples 8g-1 .. 8g+8; T % 32 == 0 */
    int dilF;                 /* wino = 3 | 4 only: the dilation of the 5x3 layer that will read V */
    float mul;                /* (ABI 14) optional scalar on top of scale[b,c]: h = act(x * scale[b,c] * mul); 0 = none.  The reverse sweep reads a gradient
                                 tensor THROUGH a pending scaled copy (dL/dres = c * dL/dy) instead of materialising the copy first */
} aid_scale_act_params;
int aid_scale_act(const aid_scale_act_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * aid_add2 -- y = a*u + b*v on [B,C,F,T] views (the `(x + res)/sqrt(2)` skip combine of a ResnetBlock whose
 *   input and output widths agree, unet...py:491 with res_conv = Identity; also gradient accumulation).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view u, v, y;
    int B, C, F, T;
    float a, b;
} aid_add2_params;
int aid_add2(const aid_add2_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Training step: parameter gradients, optimiser, EMA.
 *   replaces: torch.autograd through the network inside EDM.loss_fn (diff_params/edm.py:166-193), torch.optim.Adam.step,
 *             clip_grad_norm_ and Trainer.update_ema (training/trainer.py:253-304).
 * The activation gradients are those of the input-VJP plan (aid_conv2d on transposed weights, aid_norm_bwd, ...); the entry
 * points below add the parameter gradients.  All reductions have a fixed order (no atomics).
 *
 * aid_conv2d_wgrad : P[(b*S+s)][co][tap][ci] = alpha * sum_{f in split s, t} gy[b,co,f,t] * x[b,ci,f+(kh-KH/2)*dilF,t+kw-KW/2]
 *                    (zero padding; tap = kh*KW+kw) -- per-(sample, row split) partial weight gradients on fp32 MFMA.
 *                    `x` is the conv's input as the forward saw it (for an activated layer: the aid_scale_act output).
 * aid_wgrad_reduce : dW[co,ci,tap] (+)= sum_b gate[b,co] * in_scale[b,ci] * sum_s P      (gate / in_scale NULL -> 1)
 *                    dgate[b,co]    = sum_{ci,tap} W[co,ci,tap] * in_scale[b,ci] * sum_s P   (= alpha * <gy, ungated conv output>)
 *                    W, dW in the parameter's own layout [Cout][Cin][KH*KW].
 * aid_channel_dot  : out[b,c] = sum_{f,t} u*v
 * aid_relpos_bwd   : dW[k][h] (+)= sum_b sum_{(n,m): bucket[n][m] = k} dS[b][h][n][m] -- gradient of the relative-position embedding
 *                    [num_buckets, heads] (attention_dict.use_rel_pos; RelativePositionBias, unet...py:266-312) from the logit gradients
 *                    aid_time_attention_bwd leaves in its scratch `ws`
 * aid_scale_bwd    : scale[b,c] = gamma[c] (1 + mod[b,c]) inv[b,g]:  ds = S/scale;  dgamma[c] (+)= sum_b ds (1+mod) inv;
 *                    dmod[b,c] = ds gamma inv          (stats: [B, groups, 2] (mean, inv) of aid_group_stats)
 * aid_modulation_bwd / aid_embed_bwd : backward of aid_modulation / aid_embed (parameter gradients + d emb)
 * aid_adam         : torch.optim.Adam update over a flat buffer (bias1 = 1-beta1^t, bias2_sqrt = sqrt(1-beta2^t), gscale: optional
 *                    device scalar multiplying the gradient = the clipping coefficient written by aid_sumsq)
 * aid_ema          : dst = dst*rate + src*(1-rate)
 * aid_sumsq        : out[0] = ||x||_2, out[1] = min(1, max_norm / (||x|| + 1e-6)) (1 when max_norm <= 0); ws: AID_SUMSQ_BLOCKS doubles
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    aid_view gy, x;
    float* P;
    int B, Cin, Cout, F, T, KH, KW, dilF;
    int S;
    float alpha;
    int wino;                  /* 1 (5x3 only, T % 16 == 0): gy and x are Winograd-domain tensors [.,.,F,6,T/4] -- gy = (A gy) from aid_wino_gy, x = (B^T d) from
                                  aid_scale_act(wino=1) -- and P holds the 30 U-domain taps (xi*5+kh): P[(b*S+s)][co][30][ci]; half the MFMAs of the direct form.
                                  aid_wgrad_reduce(wino=1) applies G^T. */
} aid_conv2d_wgrad_params;
int aid_conv2d_wgrad(const aid_conv2d_wgrad_params* p, void* stream);
/* workgroups per (sample, split) of that launch (tile shape differs between the direct, the F(4,3) and the 1x1 kernels): S is sized with it */
int aid_conv2d_wgrad_tiles(int Cin, int Cout, int KH, int KW, int wino);

/* aid_wino_gy: out[b,c,f,xi,g] = (A gy)[xi] of the four samples 4g..4g+3 (A = transpose of the F(4,3) output transform): the output-gradient operand of
 * the Winograd-form weight gradient.  out rows [6][T/4]. */
typedef struct { aid_view gy, out; int B, C, F, T; } aid_wino_gy_params;
int aid_wino_gy(const aid_wino_gy_params* p, void* stream);

/* aid_pack_conv_weight: every kernel-side layout of one conv weight in one launch (replaces the torch pack ops of
 * network.prepare() after each optimiser step; Conv2d weights of unet...py:79-88 in the state_dict's own layout [Cout,Cin,KH,KW]).
 *   wp  [KH*KW][Cin_pad][Cout_pad]         wp[t][ci][co] = w[co][ci][kh][kw], t = kh*KW + kw, zero padded
 *   wpT [KH*KW][Cin_padT][Cout_padT]       the input-gradient operator: taps flipped, channel roles swapped (pack dims of (Cout, Cin)); NULL: skip
 *   wpw / wpwT [30][...]                   F(4,3) packs U = G w of both (5x3 only; fp64 arithmetic, rounded once); NULL: skip
 *   wpw8 / wpw8T [50][...]                 F(8,3) packs of both (5x3 only; G of csrc/aid_wino8.h); NULL: skip
 *   wpw2 / wpw2T [48][...]                 F(4,5) x F(4,3) packs of both (5x3 only; csrc/aid_wino45.h); NULL: skip
 *   wpw3 / wpw3T [80][...]                 F(4,5) x F(8,3) packs of both (5x3 only; rows: aid_wino45.h, samples: aid_wino8.h); NULL: skip */
typedef struct {
    const float* w;
    float* wp; float* wpT; float* wpw; float* wpwT;
    int Cout, Cin, KH, KW;
    int Cin_pad, Cout_pad, Cin_padT, Cout_padT;
    float* wpw8; float* wpw8T;
    float* wpw2; float* wpw2T;   /* 2-D packs U[xf*6 + xt] = GF w GT^T of both operators, [48][...] (5x3 only; G of csrc/aid_wino45.h); NULL: skip */
    float* wpw3; float* wpw3T;   /* 2-D packs with F(8,3) along T, U[xf*10 + xt] = GF w G8^T, [80][...] (ABI 14); NULL: skip */
} aid_pack_conv_weight_params;
int aid_pack_conv_weight(const aid_pack_conv_weight_params* p, void* stream);

typedef struct {
    const float* P; const float* W;
    const float* gate; int64_t gate_ld;
    const float* in_scale; int64_t in_scale_ld;
    float* dW; float* dgate; int64_t dgate_ld;
    int B, S, Cout, Cin, K, accumulate;
    int wino;                  /* 1: P holds 30 U-domain taps per (co, ci) (aid_conv2d_wgrad wino=1); K stays 15: dW = G^T dU */
    const float* Uw;           /* wino + dgate: the layer's F(4,3) weight pack [30][Cin_pad][Cout_pad] (<W, G^T dU> = <G W, dU>) */
    int Cin_pad, Cout_pad;
} aid_wgrad_reduce_params;
int aid_wgrad_reduce(const aid_wgrad_reduce_params* p, void* stream);

typedef struct {
    aid_view u, v;
    float* out; int64_t out_ld;
    int B, C, F, T;
} aid_channel_dot_params;
int aid_channel_dot(const aid_channel_dot_params* p, void* stream);

typedef struct {
    const float* dS;              /* [B, H, T, T]: aid_time_attention_bwd's scratch after the call */
    const int* bucket;            /* [T, T] bucket index of (query n, key m) */
    float* dW;                    /* [num_buckets, H] */
    int B, H, T, num_buckets, accumulate;
} aid_relpos_bwd_params;
int aid_relpos_bwd(const aid_relpos_bwd_params* p, void* stream);

typedef struct {
    const float* S; int64_t S_ld;
    const float* scale; int64_t scale_ld;
    const float* gamma; const float* mod; int64_t mod_ld;
    const float* stats;
    float* dgamma; float* dmod; int64_t dmod_ld;
    int B, C, groups, accumulate;
} aid_scale_bwd_params;
int aid_scale_bwd(const aid_scale_bwd_params* p, void* stream);

typedef struct {
    const float* dmod; const float* emb; const float* W;
    float* dW; float* dbias; float* demb;
    int B, E, N, accumulate;
    float* part; int64_t part_floats;   /* scratch: B * ceil(N/128) * E floats (partial sums of demb, folded in a fixed order) */
} aid_modulation_bwd_params;
int aid_modulation_bwd(const aid_modulation_bwd_params* p, void* stream);

typedef struct {
    aid_embed_params fwd;              /* the forward's parameters (emb = its output) */
    const float* demb;                 /* [B, E] */
    float* dw0; float* db0; float* dw1; float* db1; float* dw2; float* db2;
    int accumulate;
} aid_embed_bwd_params;
int aid_embed_bwd(const aid_embed_bwd_params* p, void* stream);

typedef struct {
    float* param; const float* grad; float* m; float* v;
    const float* gscale;
    int64_t n;
    float lr, beta1, beta2, eps, bias1, bias2_sqrt;
} aid_adam_params;
int aid_adam(const aid_adam_params* p, void* stream);

typedef struct { float* dst; const float* src; int64_t n; float rate; } aid_ema_params;
int aid_ema(const aid_ema_params* p, void* stream);

#define AID_SUMSQ_BLOCKS 512
typedef struct { const float* x; double* ws; float* out; int64_t n; float max_norm; } aid_sumsq_params;
int aid_sumsq(const aid_sumsq_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AID_KERNELS_H */
