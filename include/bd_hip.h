/*
 * bd_hip.h -- C ABI of libbd_hip.so, the MI355X (gfx950) kernels behind the BadDiffusion hot path.
 *
 * The reference (IBM/BadDiffusion) has no FFI for this path: the seam is the Python API that
 * baddiffusion.py calls.  Each entry point below names the reference code it replaces
 * (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator on the Python side);
 *     the library never allocates, frees or retains device memory;
 *   - activations are NHWC ("pixels x channels", fp32) with an explicit leading dimension `ld`
 *     (elements between consecutive pixels), so a channel slice of a wider buffer is a valid tensor
 *     and the reference's torch.cat on the channel axis needs no copy;
 *   - conv weights are [Cout][kh][kw][Cin] (the physical layout of a channels_last OIHW tensor);
 *   - all work is enqueued on `stream` (a hipStream_t); no hidden synchronisation, no default-stream
 *     use => capturable in a hipGraph;
 *   - return 0 on success, a negative bd_status otherwise; bd_last_error() gives the message
 *     (thread-local).  No C++ exception crosses this boundary.
 */
#ifndef BD_HIP_H
#define BD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bd_stream_t; /* hipStream_t */

enum bd_status {
    BD_OK = 0,
    BD_ERR_INVALID = -1,     /* bad shape / null pointer / misaligned */
    BD_ERR_UNSUPPORTED = -2, /* configuration the reference supports but this path does not */
    BD_ERR_LAUNCH = -3,      /* hipGetLastError() after a launch */
    BD_ERR_WORKSPACE = -4    /* workspace too small */
};

const char* bd_last_error(void);
int bd_version(void);

/* ------------------------------------------------------------------------------------------------
 * a-1 + a-2: poisoned-sample blend + BadDiffusion forward process, one fused kernel.
 * Replaces dataset.py:275-276 (get_mask), :288-315 (clean/backdoor transforms: R = m*x + (1-m)*g,
 * x0 = y | R = 0, x0 = x), util.py:83-111 (normalize, eps 1e-5) when the images are uint8, and
 * loss.py:257-285 (q_sample_diffuser) + schedulers/scheduling_ddpm.py:422-443 (add_noise):
 *     x_noisy = sqrt(ac_t) x0 + sqrt(1-ac_t) eps + (1 - sqrt(ac_t)) R,  target = rho_t R + eps.
 * Layout: NCHW in (what the reference's DataLoader yields), NHWC out with leading dim ld_out.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, C, H, W;
    const float* images_f32;      /* [B,C,H,W] normalised images, or NULL                     */
    const uint8_t* images_u8;     /* [B,H,W,C] raw uint8 (normalised in-kernel), or NULL      */
    const uint8_t* is_poison;     /* [B] 0 = clean row, 1 = backdoor row                      */
    const float* trigger;         /* [C,H,W]  g                                               */
    const float* target_img;      /* [C,H,W]  y                                               */
    const float* noise;           /* [B,C,H,W] eps (NCHW, as torch.randn produced it)         */
    const int64_t* timesteps;     /* [B]                                                      */
    const float* alphas;          /* [T]                                                      */
    const float* alphas_cumprod;  /* [T]                                                      */
    float vmin;                   /* mask threshold: m = (g > vmin) ? 0 : 1                   */
    float* x_noisy; int64_t ld_noisy;   /* NHWC [B*H*W, ld]                                   */
    float* target;  int64_t ld_target;  /* NHWC [B*H*W, ld]                                   */
    float* R_out;                 /* optional NCHW [B,C,H,W] (pixel_values), may be NULL      */
    float* x0_out;                /* optional NCHW [B,C,H,W] (target), may be NULL            */
    int64_t* mask_out;            /* optional [C,H,W] int64 mask (bit-exact check), may be NULL */
    float* image_out;             /* optional NCHW [B,C,H,W] normalised image x, may be NULL  */
    const int64_t* row_index;     /* optional [B]: batch row b reads image row_index[b] of the (larger, HBM-resident)
                                     image array -- the DataLoader's shuffled gather, fused; NULL = row b          */
    const uint8_t* flip;          /* optional [B]: 1 = mirror the image along W (RandomHorizontalFlip, dataset.py:127) */
} bd_poison_qsample_desc;
int bd_poison_qsample(const bd_poison_qsample_desc* d, bd_stream_t stream);

/* Plain q_sample_diffuser on already-collated (x0, R) NCHW batches (loss.py:257-285). */
typedef struct {
    int B, C, H, W;
    const float* x0; const float* R; const float* noise;   /* NCHW */
    const int64_t* timesteps; const float* alphas; const float* alphas_cumprod;
    float* x_noisy; int64_t ld_noisy;   /* NHWC */
    float* target;  int64_t ld_target;  /* NHWC */
} bd_qsample_desc;
int bd_qsample(const bd_qsample_desc* d, bd_stream_t stream);

/* NCHW <-> NHWC(ld) conversion for the 3-channel boundary tensors. */
int bd_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int64_t ld, bd_stream_t stream);
int bd_nhwc_to_nchw(const float* src, int64_t ld, float* dst, int B, int C, int H, int W, bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a-5 / a-6 / a-7: scheduler steps and image conversion (all NCHW-agnostic: elementwise over
 * [B, n] with per-launch scalar coefficients read from a DEVICE table indexed by t).
 * Replaces scheduling_ddpm.py:324-420, scheduling_ddim.py:261-381, pipeline_ddpm.py:115-116.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n;                    /* total elements                                            */
    const float* model_output; const float* sample; const float* noise;  /* noise may be NULL iff t == 0 */
    float* prev_sample; float* pred_original; /* pred_original may be NULL                     */
    const float* alphas_cumprod;  /* [T] device                                                */
    int t, prev_t;                /* prev_t < 0 => alpha_prod_prev = 1                         */
    int variance_type;            /* 0 fixed_small, 1 fixed_large                              */
    int clip_sample; float clip_sample_range;
    int clip_defense; float clip_defense_range;
} bd_ddpm_step_desc;
int bd_ddpm_step(const bd_ddpm_step_desc* d, bd_stream_t stream);

typedef struct {
    int64_t n;
    const float* model_output; const float* sample; const float* noise; /* noise NULL iff eta == 0 */
    float* prev_sample; float* pred_original;
    const float* alphas_cumprod;
    int t, prev_t; float final_alpha_cumprod; float eta;
    int clip_sample; float clip_sample_range;
} bd_ddim_step_desc;
int bd_ddim_step(const bd_ddim_step_desc* d, bd_stream_t stream);

/* f-4: out = clamp?(sum_j coeffs[j] * terms[j]) over k <= BD_LINCOMB_MAX dense fp32 tensors of n elements (one common
 * layout; `terms` / `coeffs` are HOST arrays read at call time).  All PRK / PLMS updates of PNDMScheduler
 * (scheduling_pndm.py:236-397: running sample, stored model outputs, Runge-Kutta accumulator; formula (9) in
 * _get_prev_sample) are such combinations, and the post-step clip of pipeline_pndm.py:108-109 is the clamp. */
#define BD_LINCOMB_MAX 6
int bd_lincomb(int k, const float* const* terms, const float* coeffs, int64_t n, int clip, float clip_range, float* out,
               bd_stream_t stream);

/* f-4: Adversarial Neuron Pruning (anp_model.py:490-514 PerturbConv2d; anp_util.py:60-88 convert_model; anp_defense.py:147 the loss that is
 * maximised).  Every Conv2d of the network is followed by a per-output-channel affine map y_c <- w_c y_c + b_c (an eval-mode batch norm with mean 0,
 * variance 1, eps 0); it commutes with the convolution, so the perturbed network is bd_unet_forward on EFFECTIVE parameters
 *   W'_c = w_c W_c,  b'_c = w_c b_c + b_c(perturbation)
 * and the perturbation's gradient is a row-wise contraction of bd_unet_backward's ordinary weight gradient:
 *   dL/dw_c = sum_k dL/dW'_ck W_ck + dL/db'_c b_c,   dL/db_c = dL/db'_c.
 * items: DEVICE array [n_items][5] int64 = (weight offset, bias offset or -1, Cout, row length = elements per output channel, offset of the
 * layer's channels in pert_w / pert_b / grad_w / grad_b); total_rows = sum of Cout.  bd_anp_apply first copies params -> eff (all other
 * tensors unchanged), then rewrites the conv rows and biases. */
int bd_anp_apply(const float* params, int64_t nparams, const float* pert_w, const float* pert_b, const int64_t* items, int n_items,
                 int64_t total_rows, float* eff, bd_stream_t stream);
/* row_norm (optional, needs pert_w): |w_c| * ||(dL/dW'_c, dL/db'_c)||, the gradient norm of the layer's own weight row + bias in the perturbed
 * network.  PerturbConv2d's conv weights are fresh trainable Parameters (anp_model.py:492-505; anp_util.freeze ran before the wrap), so the
 * reference's clip_grad_norm_(model.parameters(), 1.0) (anp_defense.py:152) takes its coefficient from bn AND conv gradients. */
int bd_anp_grad(const float* params, const float* grad_eff, const int64_t* items, int n_items, int64_t total_rows, const float* pert_w,
                float* grad_w, float* grad_b, float* row_norm, bd_stream_t stream);

/* (x/2+0.5).clamp(0,1): NCHW or NHWC(ld) in -> NHWC float [B,H,W,C] and/or uint8 round(255 x). */
int bd_to_image(const float* x, int src_is_nhwc, int64_t ld, int B, int C, int H, int W,
                float* out_f32, uint8_t* out_u8, bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a-4a: sinusoidal timestep embedding (models/embeddings.py:22-62).  t is int64 [B] (or one value
 * broadcast when t_stride == 0).
 * ------------------------------------------------------------------------------------------------ */
int bd_timestep_embedding(const int64_t* t, int t_stride, int B, int dim, int flip_sin_to_cos,
                          float freq_shift, float* out, bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) over NHWC(ld).  Replaces nn.GroupNorm + F.silu at resnet.py:559,591,
 * attention.py:125, unet_2d.py:312-313 and their autograd backward.
 * Workspace: bd_gn_workspace_bytes(B, C).  mean/rstd are [B, G] fp32 outputs saved for backward.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, HW, C, G; float eps; int silu;
    const float* x; int64_t ldx;
    const float* gamma; const float* beta;
    float* y; int64_t ldy;
    float* mean; float* rstd;       /* [B,G] */
    void* workspace; size_t workspace_bytes;
    uint16_t* y_split; int64_t ldys;   /* optional: y also / only (y == NULL) as split planes [B*HW, ldys], see
                                          "Pre-split operands" below: the operand format of bd_conv3x3_ps */
    const double* stats; int stats_splits;   /* optional (round 4): [B][stats_splits][G][2] partial (sum, sum of squares) of x written by the
                                          producer of x (bd_conv3x3_ps gn_part); read only when bd_gn_fwd_takes_stats(): the statistics
                                          pass over x is skipped                                                                     */
} bd_gn_fwd_desc;
size_t bd_gn_workspace_bytes(int B, int C);
int bd_gn_fwd_takes_stats(int B, int HW, int C, int G);   /* 1: this shape runs the statistics / apply passes (images too large for the one-pass kernels) */
int bd_gn_fwd(const bd_gn_fwd_desc* d, bd_stream_t stream);

typedef struct {
    int B, HW, C, G; int silu;
    const float* x; int64_t ldx;            /* GN input saved from forward                     */
    const float* gamma; const float* beta;
    const float* mean; const float* rstd;
    const float* dy; int64_t lddy;          /* grad wrt the (activated) output                 */
    float* dx; int64_t lddx; int accumulate_dx;  /* dx (+)= ...                                */
    float* dgamma; float* dbeta;            /* [C], written (not accumulated)                  */
    void* workspace; size_t workspace_bytes;
    float* dx_colsum; int64_t ld_colsum;    /* optional [B, C] (row stride ld_colsum): per-sample sum
                                               over pixels of the written dx (needs accumulate_dx=0);
                                               the time-embedding gradient of resnet.py:571            */
    uint16_t* dx_split; int64_t lddxs;      /* optional: the value stored to dx -- this launch's term (+ the existing dx when
                                               accumulate_dx) (+ dx_add) -- also / only (dx == NULL) as split planes         */
    const float* dx_add; int64_t ld_add;    /* optional: dx = (term (+ dx)) + dx_add, a second gradient of the same tensor
                                               folded into the store (the identity shortcut of resnet.py:596-600)     */
    float* param_partials;                  /* optional [B][2][C], caller storage: if set and bd_gn_bwd_defers(B, HW, C, G),
                                               the per-sample partial sums of (dgamma, dbeta) are left here and dgamma /
                                               dbeta are NOT written -- fold many layers later with ONE bd_gn_bwd_params
                                               launch (51 GroupNorms per CIFAR backward, one launch per segment)      */
    int dx_split_c0, dx_split_c1;           /* optional (round 4): only channels [c0, c1) of x go to dx_split, at plane column c - c0
                                               (whole 32-channel blocks; 0, 0 = all C).  The consumer that makes a tensor's gradient
                                               final hands the producer of that tensor its dY operand ready-split.            */
} bd_gn_bwd_desc;
int bd_gn_bwd(const bd_gn_bwd_desc* d, bd_stream_t stream);
/* 1 if this shape can leave its parameter partials to the caller (round 4: every valid shape -- the single-pass kernels write them
 * directly, the large-image path from its group-finalize launch), else 0 */
int bd_gn_bwd_defers(int B, int HW, int C, int G);
typedef struct {
    const float* partials;                  /* [B][2][C] as written through bd_gn_bwd_desc.param_partials */
    int C;
    float* dgamma; float* dbeta;            /* [C], written */
} bd_gn_param_item;
/* dgamma[c] / dbeta[c] = sum over the B samples of the partials, fixed order, for n layers in one launch
 * (the reduction over the batch of aten::native_group_norm_backward's weight / bias gradients, resnet.py:559,591) */
int bd_gn_bwd_params(const bd_gn_param_item* items, int n, int B, bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM engine on f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulate).
 *     C[m,n] = out_scale * (alpha * sum_k A(m,k) B(n,k) + bias[n] + rowbias[m / rows_per_group, n]
 *                           + residual[m,n])   (+ C[m,n] if accumulate)
 * Operand kinds select how (row, k) maps to memory; see DESIGN.md "igemm".
 * Replaces aten::convolution / convolution_backward / addmm / mm / bmm / baddbmm of SURVEY 2.3.
 * ------------------------------------------------------------------------------------------------ */
/* arithmetic of the contraction:
 *   BD_MODE_F32     v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation (157 TFLOP/s roof)
 *   BD_MODE_BF16X3  every fp32 operand is split on the fly into hi + lo bf16 (x = hi + lo + O(2^-17 x)) and the
 *                   product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
 *                   ~2^-16 relative error per product, 3 MFMAs on the 2.5 PFLOP/s pipe                      */
enum bd_compute_mode { BD_MODE_F32 = 0, BD_MODE_BF16X3 = 1 };

enum bd_operand_kind {
    BD_OPK_DENSE = 0, /* KC: p[row*ld + k]            RC: p[k*ld + row]                          */
    BD_OPK_CONV = 1,  /* 3x3 gather of an NHWC tensor: KC rows = output pixels, k = tap*C + c;
                         RC rows = tap*C + c, k = output pixels (wgrad)                          */
    BD_OPK_TCONV = 2, /* KC only: rows = input pixels, k = tap*C + c over dY (dgrad, any stride) */
    BD_OPK_WGT = 3    /* RC only: rows = ci, k = tap*C + co over W[co][tap][ci]   (dgrad)        */
};
typedef struct {
    int kind;              /* bd_operand_kind                                                   */
    int kc;                /* 1: k-contiguous in memory (KC)  0: row-contiguous (RC)            */
    const float* p; int64_t ld;
    int64_t bs_outer, bs_inner;   /* batch strides (elements)                                   */
    int C, Hs, Ws, Ho, Wo, stride, pad_t, pad_l, ups;   /* conv geometry (CONV/TCONV/WGT)       */
    const uint16_t* split;   /* optional, BD_MODE_BF16X3 only: the same buffer already split into bf16 hi/lo by
                                bd_split_bf16 (blocked layout below; element e of p <-> e of split); used where the
                                engine has a loader for it (weights of conv fwd / dgrad), ignored otherwise -- p must
                                always be valid                                                               */
} bd_operand;
typedef struct {
    bd_operand A, B;
    int M, N, K;
    int batch_outer, batch_inner;            /* >= 1                                            */
    float* C; int64_t ldc; int64_t c_bs_outer, c_bs_inner;
    float alpha; float out_scale;
    const float* bias;
    const float* rowbias; int64_t ld_rowbias; int rows_per_group;
    const float* residual; int64_t ldr;      /* same batch strides as C                         */
    int accumulate;
    int ksplit;                              /* 0 = choose; >1 needs workspace                  */
    void* workspace; size_t workspace_bytes;
    int tile;                                /* 0 = choose, 128 or 64                           */
    int mode;                                /* bd_compute_mode                                 */
    float* a_colsum;                         /* optional [M]: sum_k A(m,k), fused (bias gradient of a
                                                wgrad, A = dY^T); needs DENSE row-contiguous A, batch 1 */
} bd_igemm_desc;
size_t bd_igemm_workspace_bytes(const bd_igemm_desc* d);
/* Split an fp32 buffer into bf16 hi/lo exactly as the BF16X3 kernels do on the fly: hi = bf16 RNE of x (round 6; rounds 1 - 5
 * truncated: one bit less), lo = bf16 RNE of (x - hi): x = hi + lo to 2^-18 |x|.  Blocked layout, 2n uint16: for element e, hi at out[(e/32)*64 + e%32] and lo 32 entries
 * further -- one 32-element K chunk of a row is one 128-byte line (64 B hi | 64 B lo), as wide as its fp32 source.
 * n % 32 == 0, 16-byte aligned pointers.  Run once per forward over the flat weight buffer (bd_unet_forward).   */
int bd_split_bf16(const float* src, int64_t n, uint16_t* out, bd_stream_t stream);
int bd_igemm(const bd_igemm_desc* d, bd_stream_t stream);

/* Convenience wrappers over bd_igemm (all NHWC(ld), weights [Cout][3][3][Cin]).
 * conv fwd:   y[B,Ho,Wo,Cout] = conv3x3(x[B,Hs,Ws,Cin] (nearest-upsampled x2 if ups), stride, pads)
 *             (+bias) (+rowbias per sample) (+residual) ; resnet.py:493,514,118,185,201-203
 * dgrad:      dx[B,Hs<<ups,Ws<<ups,Cin] (the conv's own input grid)
 * wgrad:      dw[Cout][3][3][Cin] (written) and, if db != NULL, db[Cout] = sum over pixels of dy      */
typedef struct {
    int B, Hs, Ws, Cin, Cout, stride, pad_t, pad_l, ups; /* Ho/Wo derived by the callee         */
    int Ho, Wo;
    const float* x; int64_t ldx;
    const float* w;
    const float* bias;
    const float* rowbias; int64_t ld_rowbias;   /* [B, Cout] added per sample (time embedding)  */
    const float* residual; int64_t ldr;
    float out_scale;
    float* y; int64_t ldy;
    void* workspace; size_t workspace_bytes;
    int mode;                                   /* bd_compute_mode */
    const uint16_t* w_split;                    /* optional: w already split by bd_split_bf16           */
} bd_conv3x3_fwd_desc;
int bd_conv3x3_fwd(const bd_conv3x3_fwd_desc* d, bd_stream_t stream);

typedef struct {
    int B, Hs, Ws, Cin, Cout, stride, pad_t, pad_l, ups, Ho, Wo;
    const float* dy; int64_t lddy;
    const float* w;
    float* dx; int64_t lddx; int accumulate;   /* dx over the (virtual, upsampled) input grid   */
    void* workspace; size_t workspace_bytes;
    int mode;                                   /* bd_compute_mode */
    const uint16_t* w_split;                    /* optional: w already split by bd_split_bf16           */
} bd_conv3x3_dgrad_desc;
int bd_conv3x3_dgrad(const bd_conv3x3_dgrad_desc* d, bd_stream_t stream);

typedef struct {
    int B, Hs, Ws, Cin, Cout, stride, pad_t, pad_l, ups, Ho, Wo;
    const float* x; int64_t ldx;
    const float* dy; int64_t lddy;
    float* dw;
    void* workspace; size_t workspace_bytes;
    int mode;                                   /* bd_compute_mode */
    float* db;                                  /* optional [Cout]: bias gradient sum_pixels dy, fused  */
} bd_conv3x3_wgrad_desc;
int bd_conv3x3_wgrad(const bd_conv3x3_wgrad_desc* d, bd_stream_t stream);
size_t bd_conv3x3_workspace_bytes(int B, int Ho, int Wo, int Hs, int Ws, int Cin, int Cout, int ups);

/* ------------------------------------------------------------------------------------------------
 * Pre-split operands ("split planes"): an activation / gradient / weight matrix [rows, C] whose every
 * 32-channel block of a row is one 128-byte line, 64 B of bf16 hi then 64 B of bf16 lo (hi = bf16
 * RNE, lo = bf16 RNE of the remainder: exactly what BD_MODE_BF16X3 computes on the fly), i.e. the
 * bd_split_bf16 layout with a leading dimension: element (r, c) lives at uint16 index
 *     2*r*ld + (c/32)*64 + plane*32 + c%32          (ld % 32 == 0, 128-byte aligned base).
 * Same 4 bytes per element as fp32.  Producers: bd_split_rows (from fp32), bd_gn_fwd / bd_gn_bwd
 * (y_split / dx_split), bd_split_wt (transposed conv weights for the data gradient).
 * bd_conv3x3_ps: stride-1 pad-1 3x3 convolution (direction +1) or its data gradient (direction -1, over
 * the transposed weight planes) with both operands streamed global -> LDS by DMA; bit-identical to
 * bd_conv3x3_fwd / bd_conv3x3_dgrad in BD_MODE_BF16X3.  Replaces aten::convolution(_backward input) of
 * resnet.py:493,514 on the large layers.  H, W powers of two; K % 32 == 0; N % 128 == 0.
 * ------------------------------------------------------------------------------------------------ */
int bd_split_rows(const float* src, int64_t ld_src, int64_t rows, int C, uint16_t* dst, int64_t ld_dst, bd_stream_t stream);
/* src [B,H,W,C] fp32 -> split planes of its nearest-neighbour x2 upsampling [B,2H,2W,C] (Upsample2D, resnet.py:126-161) */
int bd_split_rows_ups2(const float* src, int64_t ld_src, int B, int H, int W, int C, uint16_t* dst, int64_t ld_dst, bd_stream_t stream);
/* W[Cout][3][3][Cin] fp32 -> split planes of Wt[Cin][3][3][Cout] (rows = ci, k = tap*Cout + co) */
int bd_split_wt(const float* w, int Cin, int Cout, uint16_t* out, bd_stream_t stream);
typedef struct {
    int B, H, W;                    /* image grid (input == output)                                  */
    int K, N;                       /* contraction channels, output channels (fwd: Cin, Cout; dgrad: Cout, Cin) */
    int direction;                  /* +1: y[p] = sum_tap x[p + tap] W[n][tap][:]   -1: dx[p] = sum_tap dy[p - tap] Wt[n][tap][:] */
    const uint16_t* x_split; int64_t ldx;   /* [B*H*W, ldx] split planes                              */
    const uint16_t* w_split;        /* [N][9][K] split planes (bd_split_bf16 of W, or bd_split_wt)   */
    const float* bias;              /* [N] or NULL                                                   */
    const float* rowbias; int64_t ld_rowbias;   /* [B, N] per-sample bias or NULL                    */
    const float* residual; int64_t ldr;
    float out_scale;                /* y = out_scale * (conv + bias + rowbias + residual)            */
    float* y; int64_t ldy; int accumulate;
    void* workspace; size_t workspace_bytes;   /* >= bd_conv3x3_ps_workspace_bytes(): K-split slabs of the small-layer variant */
    double* gn_part; int gn_groups; /* optional (round 4): the statistics of the GroupNorm that reads y next from this call's epilogue instead
                                       of a pass over y.  The plan uses it for resnet.py:591 (norm2 behind conv1) only: a conv2 -> next
                                       block's norm1 hand-over is NOT implemented (norm1 inputs are concat buffers / residual sums):
                                       [B][S][gn_groups][2] fp64 partial (sum, sum of squares) per sample and 256-pixel tile, S =
                                       bd_conv3x3_ps_gn_splits() > 0; forward calls with exactly one of rowbias / residual, no accumulate.
                                       Hand them to bd_gn_fwd as stats / stats_splits.                                              */
} bd_conv3x3_ps_desc;
size_t bd_conv3x3_ps_workspace_bytes(const bd_conv3x3_ps_desc* d);
int bd_conv3x3_ps_gn_splits(int B, int H, int W, int K, int N, int groups);   /* 0: this call cannot write GroupNorm partials */
int bd_conv3x3_ps(const bd_conv3x3_ps_desc* d, bd_stream_t stream);
/* weight (and bias) gradient of the same convolution, both operands as split planes:
 *   dw[Cout][3][3][Cin] = sum_p dy[p][co] x[p + tap][ci],  db[Cout] = sum_p dy[p][co] (optional, same launch).
 * Cin, Cout % 128 == 0; K (pixels) is split over workgroups with a fixed-order second pass (deterministic);
 * workspace >= bd_conv3x3_ps_wgrad_workspace_bytes().  Replaces aten::convolution_backward (weight, bias).   */
typedef struct {
    int B, H, W, Cin, Cout;
    const uint16_t* x_split; int64_t ldx;     /* conv input  [B*H*W, ldx]  */
    const uint16_t* dy_split; int64_t lddy;   /* output grad [B*H*W, lddy] */
    float* dw; float* db;
    void* workspace; size_t workspace_bytes;
} bd_conv3x3_ps_wgrad_desc;
/* Run-time tuning knob for measurement sweeps (bench.py --gpus N sweeps the weight-gradient slot count inside ONE process, because a
 * multi-GPU node is leased once): key "ps_wg3_slots" = K-split slots of conv_ps_wgrad3_kernel (0 restores BD_PS_WG3_SLOTS / the default:
 * 3/4 of the CUs, 1/2 for strip-order images).  Workspace sizes depend on it: every plan lays out again on its next call, callers must
 * re-query bd_unet_workspace_bytes.  Results are unchanged up to the fp32 summation order of the K-split. */
int bd_tune_set(const char* key, int value);
size_t bd_conv3x3_ps_wgrad_workspace_bytes(const bd_conv3x3_ps_wgrad_desc* d);
int bd_conv3x3_ps_wgrad(const bd_conv3x3_ps_wgrad_desc* d, bd_stream_t stream);

/* out[g, n] = sum over rows m in group g of x[m, n]  (rows_per_group rows each); bias / temb grads. */
int bd_colsum(const float* x, int64_t ldx, int64_t rows, int N, int64_t rows_per_group, float* out,
              int64_t ld_out, int accumulate, bd_stream_t stream);
/* dx[b, y, x, c] (+)= sum of the 2x2 block of du[b, 2y.., 2x.., c]  (nearest-upsample backward). */
int bd_sum2x2(const float* du, int64_t ldu, float* dx, int64_t lddx, int B, int H, int W, int C,
              int accumulate, bd_stream_t stream);

/* Row softmax over [rows, n] (attention.py:161) and its backward dS = P * (dP - sum(dP*P)). */
int bd_softmax_fwd(const float* s, float* p, int64_t rows, int n, bd_stream_t stream);
int bd_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int n, bd_stream_t stream);

/* y = x * sigmoid(x) ; dx = dy * silu'(x)  (embeddings.py:205-206, resnet.py:576) */
int bd_silu_fwd(const float* x, float* y, int64_t n, bd_stream_t stream);
int bd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, int accumulate, bd_stream_t stream);

/* a-3: loss = mean((pred - target)^2) and dpred = 2 (pred - target) / n  (loss.py:301).
 * loss_type: 0 l2, 1 l1, 2 huber(beta=1).  loss is a device scalar, written (not accumulated).
 * grad_scale multiplies dpred (1/world for DP mean).  workspace >= bd_reduce_workspace_bytes().   */
size_t bd_reduce_workspace_bytes(void);
int bd_loss_fwd_bwd(const float* pred, int64_t ldp, const float* target, int64_t ldt, int64_t rows, int C,
                    int loss_type, float grad_scale, float* loss, float* dpred, int64_t lddp,
                    void* workspace, bd_stream_t stream);

/* f-3: mean structural similarity of two [N,C,H,W] image batches in [0, data_range] given by element strides (NCHW or
 * NHWC storage alike): StructuralSimilarityIndexMeasure(data_range=1.0) of baddiffusion.py:536-547 with the torchmetrics
 * defaults (11x11 Gaussian, sigma 1.5, k1 0.01, k2 0.03, reflect padding, cropped border, mean over C,H,W then batch).
 * out: device float scalar, written.  workspace >= bd_ssim_workspace_bytes(N, C, H, W).  Needs H, W > 10, N*C <= 65535. */
size_t bd_ssim_workspace_bytes(int N, int C, int H, int W);
int bd_ssim(const float* preds, const float* target, int N, int C, int H, int W, int64_t stride_n, int64_t stride_c,
            int64_t stride_h, int64_t stride_w, float data_range, float* out, void* workspace, size_t workspace_bytes,
            bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * FID feature extractor (measure path, SURVEY f-3): the layer kernels of pytorch_fid's InceptionV3 up to pool3
 * (/root/reference/fid_score.py:53 `from pytorch_fid.inception import InceptionV3`, :91-148 get_activations, :255
 * `InceptionV3([block_idx])`; pytorch-fid==0.2.1 per requirements.txt -- a third-party package absent from the reference tree,
 * its published graph is restated in baddiffusion_amd/inception.py).  NHWC fp32, exact fp32 products (v_mfma_f32_32x32x2_f32).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* x; int64_t ldx;      /* input  [B, H, W, Cin], pixel stride ldx floats (a channel slice of a wider buffer is fine) */
    const float* w;                   /* weights [KH][KW][Cin][Cout] (BatchNorm scale folded in by the caller)                     */
    const float* bias;                /* optional [Cout] (folded BatchNorm shift)                                                  */
    float* y; int64_t ldy;            /* output [B, Ho, Wo, Cout] at pixel stride ldy: Ho = (H + 2 pad_h - KH) / stride_h + 1, ...  */
    int B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w;
    int relu;                         /* y = max(y, 0)  (BasicConv2d = conv + bn + relu)                                           */
    int w_kc;                         /* round 5: 1 = weights are [KH][KW][Cout][Cin] (K contiguous): the double-buffered 128 x 64 tile kernel
                                         with vector LDS reads; 0 = [KH][KW][Cin][Cout], the 64 x 64 kernel.  Same results up to the fp32
                                         summation order inside a 16-channel chunk.                                                   */
} bd_conv2d_desc;
/* Cout % 4 == 0; x 16-byte aligned with ldx % 4 == 0 when Cin % 4 == 0 (any alignment for the 3-channel stem). */
int bd_conv2d_nhwc(const bd_conv2d_desc* d, bd_stream_t stream);
/* F.max_pool2d (mode 0, padding = -inf) / F.avg_pool2d (mode 1, count_include_pad as given) with a square window, C % 4 == 0. */
int bd_pool2d_nhwc(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int H, int W, int C, int kernel, int stride, int pad,
                   int mode, int count_include_pad, bd_stream_t stream);
/* y = scale * F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) + shift; x is [B, H, W, C] uint8 (read as
 * value / 255, i.e. after ToTensor) when x_is_u8, else float32. */
int bd_resize_bilinear_nhwc(const void* x, int x_is_u8, float* y, int B, int H, int W, int C, int Ho, int Wo, float scale, float shift,
                            bd_stream_t stream);
/* AdaptiveAvgPool2d((1, 1)): y[b, c] = mean over the HW pixels of x [B, HW, C] (pixel stride ldx), fixed summation order. */
int bd_global_avgpool_nhwc(const float* x, int64_t ldx, float* y, int B, int HW, int C, bd_stream_t stream);


/* a-8: global-norm clip + Adam over one flat fp32 buffer (baddiffusion.py:320, 611-615).
 * bd_sumsq: sumsq (device double scalar) = sum g^2.  bd_adam_clip reads it:
 *   coef = min(1, max_norm / (sqrt(sumsq) + 1e-6));  g *= coef;  Adam(b1,b2,eps), bias correction from
 *   `step` (1-based, host scalar) ; lr host scalar.  grad_norm_out (device float, optional) = sqrt(sumsq). */
int bd_sumsq(const float* g, int64_t n, double* sumsq, void* workspace, bd_stream_t stream);
int bd_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                 double max_norm, double lr, double b1, double b2, double eps, int step, float* grad_norm_out,
                 bd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a-4: the whole UNet2DModel (models/unet_2d.py:82-326) as a static plan: forward (inference or
 * training, saving what backward needs in the caller's workspace) and backward (writes the flat
 * gradient buffer; every parameter gets exactly one contribution, so no zeroing is needed).
 * Parameters live in ONE flat fp32 buffer; bd_unet_param_* describe where each state_dict key is.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int sample_size, in_channels, out_channels;
    int num_blocks;                 /* <= 8                                                     */
    int block_out_channels[8];
    int down_attn[8];               /* 1 = AttnDownBlock2D, 0 = DownBlock2D                     */
    int up_attn[8];                 /* 1 = AttnUpBlock2D,   0 = UpBlock2D                       */
    int layers_per_block;
    int downsample_padding;         /* 0 => asymmetric (0,1,0,1) pad, 1 => symmetric            */
    int flip_sin_to_cos; float freq_shift;
    float norm_eps; int norm_num_groups;
    int attention_head_dim;         /* 0 => single head                                         */
    float mid_block_scale_factor;
    int compute_mode;               /* bd_compute_mode of every conv / GEMM of the network      */
} bd_unet_config;

typedef struct bd_unet bd_unet;     /* host-only plan object (no device memory)                  */
int bd_unet_create(const bd_unet_config* cfg, bd_unet** out);
void bd_unet_destroy(bd_unet* u);
int bd_unet_set_compute_mode(bd_unet* u, int mode);   /* bd_compute_mode; may be changed between calls */
int64_t bd_unet_num_params(const bd_unet* u);
int bd_unet_num_tensors(const bd_unet* u);
/* i-th parameter tensor: state_dict key, offset (elements) in the flat buffer, logical shape
 * (rank <= 4, OIHW for convs) and layout: 0 = as-is (contiguous logical shape),
 * 1 = conv weight stored [O][kh][kw][I]. */
int bd_unet_param_info(const bd_unet* u, int i, const char** name, int64_t* offset, int* rank,
                       int64_t shape[4], int* layout);
/* workspace bytes for batch B; training != 0 keeps every tensor backward needs */
size_t bd_unet_workspace_bytes(bd_unet* u, int B, int training);
/* x: NHWC [B*S*S, ldx] ; t: int64 [B] (t_stride 0 => broadcast t[0]) ; out NHWC [B*S*S, ldo] */
int bd_unet_forward(bd_unet* u, int B, int training, const float* params, const float* x, int64_t ldx,
                    const int64_t* t, int t_stride, float* out, int64_t ldo,
                    void* workspace, size_t workspace_bytes, bd_stream_t stream);
/* after a training forward with the same workspace and the same x: dout NHWC -> grads (flat, same offsets) */
int bd_unet_backward(bd_unet* u, int B, const float* params, const float* x, int64_t ldx,
                     const float* dout, int64_t lddo, float* grads, void* workspace, size_t workspace_bytes,
                     bd_stream_t stream);
/* backward split in `bd_unet_num_segments` contiguous-in-time segments so the caller can overlap a
 * gradient all-reduce with the rest of backward: after segment s, grads in
 * [seg_lo[s], seg_hi[s]) (elements) are final. */
/* Backward runs the weight-gradient GEMMs on a second, low-priority stream owned by the plan (forked from / joined to
 * `stream` with events, once per node -- still hipGraph-capturable and free of host synchronisation).  0 disables it. */
int bd_unet_set_aux_stream(bd_unet* u, int enabled);
/* Sampling loops (pipeline_ddpm.py:106-111, pipeline_ddim.py:114-121) evaluate the network 50-1000 times over the same parameters.
 * While enabled, an inference forward whose (params pointer, workspace pointer, B) equal those of the previous inference forward
 * skips the per-forward weight preprocessing (split-plane copy of the flat buffer, pre-summed upsample tap planes).  The caller
 * promises not to modify the parameters in between; switching it on or off re-reads them once.  Training forwards never skip. */
int bd_unet_set_static_weights(bd_unet* u, int enabled);
/* Forget the prepared planes without leaving the block: call after an in-place parameter change (load_state_dict, an optimizer step on
 * the same buffer) or when the workspace was freed and may be re-allocated at the same address.  bd_unet_set_compute_mode and every
 * re-layout (another batch size / training flag) do the same internally. */
int bd_unet_reset_static_cache(bd_unet* u);
int bd_unet_num_segments(const bd_unet* u);
int bd_unet_segment_range(const bd_unet* u, int seg, int64_t* lo, int64_t* hi);   /* host-only query: the main range */
/* a segment finalises 1 or 3 ranges of the flat gradient: k = 0 its own parameters, k = 1 / 2 its resnets' rows of the
 * batched time_emb_proj weight / bias (stored with the time embedding, computed in the resnets' backward) */
int bd_unet_segment_num_ranges(const bd_unet* u, int seg);
int bd_unet_segment_range_k(const bd_unet* u, int seg, int k, int64_t* lo, int64_t* hi);
/* Deferred join (data-parallel training).  By default every bd_unet_backward_segment call ends with the caller's stream ordered
 * after the weight gradients it put on the plan's side stream.  With bd_unet_set_deferred_join(u, 1) only the LAST segment joins:
 * a segment's weight gradients keep running beside the next segment's data-gradient chain, and whoever consumes the segment's
 * gradient ranges (the all-reduce) is ordered behind them explicitly: bd_unet_stream_wait_aux(u, s) makes stream `s` wait for
 * everything the segment calls made so far have put on the side stream (no host synchronisation; a no-op before the first
 * backward or with the side stream disabled). */
int bd_unet_set_deferred_join(bd_unet* u, int enabled);
int bd_unet_stream_wait_aux(bd_unet* u, bd_stream_t stream);
int bd_unet_backward_segment(bd_unet* u, int seg, int B, const float* params, const float* x, int64_t ldx,
                             const float* dout, int64_t lddo, float* grads, void* workspace, size_t workspace_bytes,
                             bd_stream_t stream, int64_t* ready_lo, int64_t* ready_hi);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py `roofline`): when enabled, every igemm launch is bracketed by a
 * hipEvent pair on its own stream; totals are kept per kernel class ("igemm_<tile>_<A>_<B>").
 * flops = 2*M*N*K algorithmic ; bytes = unique operand + output bytes (algorithmic, fp32).
 * ------------------------------------------------------------------------------------------------ */
int bd_prof_enable(int on);
int bd_prof_enabled(void);   /* 1 while launches are being bracketed by events (such launches cannot be graph-captured) */
int bd_prof_reset(void);
int bd_prof_num_classes(void);
int bd_prof_get(int cls, const char** name, int64_t* launches, double* total_ms, double* flops, double* bytes);

/* ------------------------------------------------------------------------------------------------
 * Phase-decomposed convolutions on split planes (round 3, csrc/conv_ph.hip).
 * Upsample2D (resnet.py:126-161) is y = conv3x3(nearest_up2(x)): every output pixel's 3x3 window covers only 2x2 distinct
 * source pixels, so with the tap sums E[oy][ox] (bd_upsample_weights) the forward is four 2x2-tap convolutions on the SOURCE
 * grid (one per output pixel class), the data gradient ONE 16-tap convolution sampling dY at stride 2, and the weight gradient
 * 16 tap products folded back to the nine taps: 16 instead of 36 tap products per source pixel, same result up to the fp32
 * rounding of the weight sums.  The data gradient of a stride-2 convolution (Downsample2D, resnet.py:199-208) is the same
 * construction with 4 / 2 / 2 / 1 taps per input-pixel parity class (bd_conv3x3_s2_dgrad_ps).
 * All operands are split planes (see above): H, W powers of two, Cin, Cout multiples of 128.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, H, W, Cin, Cout;                     /* H x W = SOURCE grid; the convolution runs on 2H x 2W            */
    const uint16_t* x_split; int64_t ldx;       /* source activations [B*H*W, Cin]                 (fwd, wgrad)    */
    const uint16_t* dy_split; int64_t lddy;     /* output gradient on the fine grid [B*4HW, Cout]  (dgrad, wgrad)  */
    const uint16_t* e_split;                    /* bd_upsample_weights: E   [Cout][16][Cin]        (fwd)           */
    const uint16_t* et_split;                   /* bd_upsample_weights: E^T [Cin][16][Cout]        (dgrad)         */
    const float* bias;                          /* fwd, optional [Cout]                                            */
    float* y; int64_t ldy;                      /* fwd out [B*4HW, Cout]                                           */
    float* dx; int64_t lddx; int accumulate;    /* dgrad out [B*HW, Cin], += when accumulate                       */
    float* dw; float* db;                       /* wgrad out [Cout,3,3,Cin] and (optional) [Cout]                  */
    void* workspace; size_t workspace_bytes;    /* wgrad / dgrad: bd_upsample_conv_{wgrad,dgrad}_workspace_bytes   */
} bd_upsample_conv_desc;
int bd_upsample_weights(const float* w /* [Cout,3,3,Cin] */, int Cin, int Cout, uint16_t* e_split, uint16_t* et_split /* may be null: forward only */,
                        bd_stream_t stream);
int bd_upsample_conv_fwd(const bd_upsample_conv_desc* d, bd_stream_t stream);
int bd_upsample_conv_dgrad(const bd_upsample_conv_desc* d, bd_stream_t stream);
int bd_upsample_conv_wgrad(const bd_upsample_conv_desc* d, bd_stream_t stream);
size_t bd_upsample_conv_wgrad_workspace_bytes(const bd_upsample_conv_desc* d);
size_t bd_upsample_conv_dgrad_workspace_bytes(const bd_upsample_conv_desc* d);   /* 0 unless the grid is small (taps dealt to 4 workgroups per tile) */

typedef struct {
    int B, Ho, Wo, Cin, Cout, pad;              /* stride-2 conv: input grid 2Ho x 2Wo; pad 0 = F.pad(0,1,0,1) + padding 0, 1 = padding 1 */
    const uint16_t* dy_split; int64_t lddy;     /* [B*Ho*Wo, Cout] */
    const uint16_t* wT_split;                   /* transposed weight planes Wt[ci][9][co] (bd_split_wt) */
    float* dx; int64_t lddx; int accumulate;    /* [B*4HoWo, Cin] */
} bd_conv3x3_s2_dgrad_desc;
int bd_conv3x3_s2_dgrad_ps(const bd_conv3x3_s2_dgrad_desc* d, bd_stream_t stream);

/* Dense (batched) GEMM on split planes -- the attention block's GEMMs (attention.py:85-186: query / key / value / proj_attn Linear
 * layers, baddbmm(q, k^T), bmm(probs, v)) and their backward, without converting an operand on the way into LDS:
 *     C[m][n] = out_scale * (alpha * sum_k A[m][k] B[n][k] + bias[n] + residual[m][n])   (+ C[m][n] when accumulate)
 * a / b are split planes (bd_split_rows layout; ld and batch strides in 4-byte units, as for the fp32 tensor they replace):
 *   K-contiguous (x_kmajor = 0): rows = M (a) or N (b, "weights [N][K]"), K along the row;
 *   K-major      (x_kmajor = 1): rows = K, the M (a) or N (b) index along the row (both operands of a weight gradient, V of P V).
 * The result is written as fp32 (c), as split planes (c_split: the next GEMM's operand), or both.  a_colsum (optional, both operands
 * K-major, no batch): a_colsum[m] = sum_k A[k][m], the bias gradient of a Linear layer, from the same launch.
 * M, N % 128 == 0, K % 32 == 0; operand planes 128-byte aligned.  Long-K products with few tiles split K over workgroups with a
 * fixed-order second pass (deterministic): workspace >= bd_gemm_sp_workspace_bytes().                                           */
typedef struct {
    int M, N, K, batch;                          /* batch <= 1: one product                                        */
    const uint16_t* a; int64_t lda, a_bs; int a_kmajor;
    const uint16_t* b; int64_t ldb, b_bs; int b_kmajor;
    float* c; int64_t ldc, c_bs;                 /* fp32 output [batch][M][ldc] or NULL                            */
    uint16_t* c_split; int64_t ldcs, cs_bs;      /* split-plane output (rows of ldcs 4-byte units) or NULL         */
    const float* bias;                           /* [N] or NULL                                                    */
    const float* residual; int64_t ldr, r_bs;    /* [batch][M][ldr] or NULL                                        */
    float alpha, out_scale;
    int accumulate;                              /* c += ...  (needs c)                                            */
    float* a_colsum;                             /* [M] or NULL                                                    */
    void* workspace; size_t workspace_bytes;
} bd_gemm_sp_desc;
size_t bd_gemm_sp_workspace_bytes(const bd_gemm_sp_desc* d);
int bd_gemm_sp(const bd_gemm_sp_desc* d, bd_stream_t stream);

/* Attention core on split planes (attention.py:148-162 and its backward), N = 256 tokens, head dim 256:
 *   forward : o = softmax(scale * q k^T) v                         [+ pt_split = P^T planes, needed by the backward]
 *   backward: dq = scale * dS k, dk = scale * dS^T q, dv = P^T dO,  dS = P o (dP - rowsum(P o dP)), dP = dO v^T
 * qkv_split: planes of the QKV projection's output [B*N, ld] (q | k | v column blocks, head h at column h*dh of each);
 * o_split / do_split [B*N, C]; pt_split / dst_split [B*heads, N, N] planes of the TRANSPOSED probability / score-gradient matrices
 * (rows = keys); dqkv_split: planes in the layout of qkv_split.  No [N, N] fp32 matrix is written.  The [N, N] products, the softmax
 * and their backward run in three launches per block instead of eight.  Other shapes: BD_ERR_UNSUPPORTED (bd_attn_sp_supported). */
typedef struct {
    int B, heads, N, dh;
    const uint16_t* qkv_split; int64_t ld;
    float scale;
    uint16_t* o_split; int64_t ldo;            /* forward out */
    uint16_t* pt_split;                        /* forward out (NULL: inference), backward in */
    const uint16_t* do_split; int64_t lddo;    /* backward in */
    uint16_t* dst_split;                       /* backward scratch [B*heads, N, N] planes */
    uint16_t* dqkv_split; int64_t lddqkv;      /* backward out */
} bd_attn_sp_desc;
int bd_attn_sp_supported(int N, int dh);
int bd_attn_sp_fwd(const bd_attn_sp_desc* d, bd_stream_t stream);
int bd_attn_sp_bwd(const bd_attn_sp_desc* d, bd_stream_t stream);

/* What the matrix pipe alone sustains on THIS board, now: v_mfma_f32_32x32x16_bf16 on register operands only (no memory
 * traffic), 2 workgroups of 512 threads per CU, `iters` x 12 MFMAs per wave, launched `launches` times back to back and timed
 * with a hipEvent pair on `stream` (blocks until done).  random_operands = 0: small constant integers (data-independent
 * issue rate, ~2.47 PFLOP/s); 1: per-lane random bf16 in [-1, 1) -- the board's power limit pulls the clock down and the
 * figure is what bounds a real kernel (bench.py `roofline.mfma_power_limited_peak_measured`).  *tflops = bf16 MFMA TFLOP/s. */
int bd_mfma_probe(int random_operands, int iters, int launches, double* tflops, bd_stream_t stream);

/* dst[i] (+)= scale * src[i] over a flat fp32 range (gradient accumulation across micro-batches). */
int bd_axpy(const float* src, float* dst, int64_t n, float scale, int accumulate, bd_stream_t stream);

/* bd_adam_clip with the step-dependent scalars read from DEVICE memory (hipGraph replay):
 * hyper = { lr / (1 - b1^step), sqrt(1 - b2^step) } */
int bd_adam_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                     double max_norm, const float* hyper, double b1, double b2, double eps,
                     float* grad_norm_out, bd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BD_HIP_H */
