/* otvm_hip.h -- C ABI of libotvm_hip.so: the MI355X (gfx950) kernels behind the OTVM per-frame
 * inference path.
 *
 * The reference (Hongje/OTVM) has no FFI: its drop-in boundary is the Python nn.Module call
 * `EvalModel.forward` (reference models/alpha/model.py:391-512).  The product keeps that Python
 * surface (otvm_amd/alpha_model.py) and implements every device operation below it through this
 * C ABI -- plain pointers, sizes and a hipStream_t; no torch types.  Each entry point cites the
 * reference code whose device work it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless noted; `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC fp32 with an explicit pixel stride `ld` (elements), so a tensor may be
 *     a channel slice of a wider buffer (this is how every torch.cat of the reference disappears);
 *   - channel counts of device tensors are padded to a multiple of 4 (zero weights on the pad);
 *   - every function returns 0 on success, non-zero on error; otvm_last_error() describes it.
 */
#ifndef OTVM_HIP_H
#define OTVM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTVM_ACT_NONE 0
#define OTVM_ACT_RELU 1
#define OTVM_ACT_LEAKY 2 /* negative slope 0.01 (nn.LeakyReLU default, FBA/models.py:305) */

/* Convolution arithmetic.  F32 and F16X3 accumulate in fp32 and meet the reference's 1e-3 fp32 contract:
 *   F32   : v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak);
 *   F16X3 : each fp32 operand split into fp16 hi+lo (22 significant bits), three
 *           v_mfma_f32_32x32x16_f16 passes hi*hi + hi*lo + lo*hi (833 TFLOP/s fp32-equivalent peak);
 *   F16   : (ABI 18) a LABELLED reduced-precision mode, never a default and not covered by the parity contract: operands rounded
 *           to fp16 once, ONE v_mfma_f32_32x32x16_f16 pass, fp32 accumulate -- in the implicit-GEMM and 3x3 patch kernels; the
 *           other f16x3 kernels (stems, 16-wide head tile, fused STM bottleneck, memory read) keep their three passes.  Takes
 *           the f16x3 weight arrays (w_hi / w_scale / w_frag / w_wfrag).                                                   */
#define OTVM_PREC_F32 0
#define OTVM_PREC_F16X3 1
#define OTVM_PREC_F16 2

const char* otvm_last_error(void);
#define OTVM_ABI_VERSION 19   /* 2: otvm_ppm_pool_ws_bytes(H, C); 3: otvm_conv_params.in_scale/in_shift/in_act;
                                 4: otvm_conv_params.splitk_ws; 5: otvm_preprocess_params.fg_u8/bg_u8/u8_rgb;
                                 6: otvm_conv_params.tune + otvm_conv2d_candidates;
                                 7: folded GroupNorm tables on otvm_gn_apply's residual and otvm_upsample_bilinear's input;
                                 8: otvm_memory_read_f16x3_partial / _combine / _partial_count; 9: otvm_ppm_head;
                                 10: otvm_finite_guard, otvm_clear;
                                 11: batch of images per launch (otvm_conv_params.batch ..., otvm_gn_*_b, otvm_upsample_bilinear_b,
                                     otvm_maxpool3x3s2_b);
                                 12: training forward (otvm_fba_head_train, otvm_upsample4_logits3, otvm_trimap_to_sm, otvm_loss_*);
                                 13: otvm_conv_params.w_wfrag + otvm_pack_wave_weight_f16x3 (one-wave 64x64 tile);
                                 14: otvm_ppm_conv_z / otvm_ppm_conv_add (the PPM branches' share of conv_up1.0 without upsampling);
                                 15: otvm_stm_bottleneck_f16x3 (one kernel per 1/4-resolution bottleneck of the STM encoders);
                                 19: ... and per 1/8-resolution identity bottleneck (Cin = 512, otvm_stm_bottleneck_params.tile);
                                 16: otvm_conv_params.gn_gamma ... gn_counter (the GroupNorm scale / shift table of the OUTPUT
                                     written by the conv's last workgroup instead of a separate otvm_gn_table launch);
                                 17: otvm_gram_f16 / otvm_gn_predict (GroupNorm statistics of a 1x1 convolution's output predicted
                                     from its input's Gram matrix: the normalisation moves into that convolution's epilogue);
                                 18: otvm_gram_params.diag / otvm_gn_predict_params.diag (conditioning + saturation diagnostics of the
                                     predicted statistics); implicit-GEMM tiles 32 + t with LDS-DMA weight stages and 64 + t = the same on
                                     v_mfma_f32_16x16x32_f16 (tune codes; no new entry points) */
int otvm_abi_version(void);

/* ---------------------------------------------------------------- weights (load time) ----------
 * Pack an OIHW fp32 conv weight into the K-major layout the implicit-GEMM kernel reads:
 * w_packed[O_pad][K_pad], k = (ky*kw + kx)*I_pad + c, zero padded.  otvm_conv2d reads whole N tiles of
 * weight rows: O_pad must be O rounded up to a multiple of 128 (the same holds for w_hi / w_lo).
 *   ws != 0  : apply weight standardisation first (layers_WS.py:15-21: subtract per-filter mean,
 *              divide by sqrt(unbiased var + 1e-12) + 1e-5) -- done ONCE here instead of per forward.
 *   scale    : optional per-output-channel multiplier (BatchNorm eval fold gamma/sqrt(var+eps),
 *              torchvision Bottleneck inside STM.py:43-51,79-87), may be NULL.                    */
int otvm_pack_conv_weight(const float* w_oihw, int O, int I, int kh, int kw, int ws, const float* scale,
                          float* w_packed, int O_pad, int I_pad, int K_pad, void* stream);

/* f16x3, 3x3 stride-1 convolutions with I_pad % 32 == 0: weights in MFMA B-fragment order for the patch kernel
 * (conv_patch_f16x3.hip), [I_pad/32][9 taps][ceil(O/32)][2 k-steps][hi|lo][64 lanes][8 halfs].  w_scale as above. */
int64_t otvm_patch_weight_bytes_f16x3(int O, int I_pad);
int otvm_pack_patch_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, void* w_frag, float* w_scale,
                                 void* stream);

/* f16x3, 7x7 stride-2 stems with few input channels (I_pad <= 64, O <= 64): weights in MFMA B-fragment order for the stem
 * kernel (conv_stem_f16x3.hip), [ceil(I_pad/8) channel groups][25 k-steps of two taps][2 n-tiles][hi|lo][64 lanes][8 halfs];
 * goes into otvm_conv_params.w_frag of such a layer.                                                              */
int64_t otvm_stem_weight_bytes_f16x3(int I_pad);
int otvm_pack_stem_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, void* w_frag, float* w_scale,
                                void* stream);

/* Fold an eval-mode BatchNorm (eps 1e-5, running statistics; torchvision Bottleneck used by
 * STM.py:43-51,79-87) into a per-channel scale/bias: scale = gamma/sqrt(var+eps),
 * bias = beta - mean*scale.                                                                       */
int otvm_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int n,
                 float* scale, float* bias, void* stream);

/* ---------------------------------------------------------------- convolution ------------------
 * Implicit-GEMM convolution on the matrix cores (see OTVM_PREC_*), replaces every F.conv2d of
 * the path (180 per non-first frame).  out = act(conv(in') + bias + residual) where
 * in' = relu(in) if in_relu (STM.py:23-24 pre-activation ResBlock) else in.                       */
typedef struct {
    const float* in;  int H, W, Cin, in_ld;         /* Cin: padded channel count, multiple of 4 */
    const float* w;   int K_pad;                    /* packed weight from otvm_pack_conv_weight */
    const float* bias;                              /* [Cout] or NULL                           */
    const float* residual; int res_ld;              /* [Ho*Wo, Cout] view or NULL               */
    float* out;       int Ho, Wo, Cout, out_ld;
    int kh, kw, stride, pad, dil;
    int in_relu, act;
    int precision;                                  /* OTVM_PREC_F32 | OTVM_PREC_F16X3 | OTVM_PREC_F16 */
    const void* w_hi; const void* w_lo;             /* f16x3: split weights [O_pad][K_pad] fp16   */
    const float* w_scale;                           /* f16x3: per-filter power-of-two scale [Cout] */
    const void* w_frag;                             /* f16x3, optional: fragment-major weights for the 3x3 patch kernel
                                                       (otvm_pack_patch_weight_f16x3) or, for a 7x7 stride-2 layer, the
                                                       stem kernel (otvm_pack_stem_weight_f16x3); NULL = implicit GEMM only */
    double* gn_stats;                               /* optional: fused GroupNorm(32) statistics of the OUTPUT
                                                       (sum, sum of squares per group, [32][2] fp64, accumulated
                                                       atomically; Cout = 64, 128, 256, ... (32 groups of a power-of-two
                                                       number of channels), act == NONE, no residual) or NULL        */
    const float* in_scale; const float* in_shift;   /* optional fused normalisation of the INPUT (the GroupNorm apply of
                                                       the producing layer folded into this conv's staging):
                                                       in' = in_act(in * in_scale[c] + in_shift[c]) inside the image, zero
                                                       padding outside; tables of Cin floats from otvm_gn_table, or NULL.
                                                       Only layers for which otvm_conv2d_accepts_input_norm() returns 1 */
    int in_act;                                     /* OTVM_ACT_* applied after the input normalisation */
    int tune;                                       /* 0 = built-in heuristic; else a configuration code returned by
                                                       otvm_conv2d_candidates (which kernel / tile / K split runs the layer) */
    void* splitk_ws; int64_t splitk_ws_bytes;       /* optional workspace (f16x3): layers with too few output tiles to fill
                                                       the chip split K over up to 8 workgroups per tile and reduce the
                                                       partial tiles in a fixed order (deterministic); NULL = never split.
                                                       One workspace per stream that runs convs concurrently.          */
    /* ---- batch (ABI 11): the same layer applied to `batch` images in ONE launch (independent video sequences stepped
     * in lock-step share the weights; small maps cannot fill 256 CUs from one image).  Image b of a tensor lives `*_bs`
     * ELEMENTS behind image 0; every image is computed exactly as a batch-1 launch computes it (same tiles, same
     * summation order).  batch <= 1: a single image, the strides are ignored.                                      */
    int batch;
    int64_t in_bs, out_bs, res_bs;                  /* floats between consecutive images of in / out / residual        */
    int gn_bs;                                      /* doubles between the images' [32][2] statistics blocks (gn_stats) */
    int norm_bs;                                    /* floats between the images' in_scale / in_shift tables            */
    const void* w_wfrag;                            /* f16x3, optional (ABI 13): the split weights in MFMA B-fragment order for
                                                       the one-wave 64x64 tile (otvm_pack_wave_weight_f16x3); layers with
                                                       Cin % 32 == 0; NULL = that tile is never a candidate               */
    /* ---- ABI 16, optional, with gn_stats: the per-channel table otvm_gn_table would compute from the finished statistics
     * (scale[c] = rstd[g] gamma[c], shift[c] = beta[c] - mean[g] scale[c]; same arithmetic, same bits) is written by the
     * LAST workgroup of the launch to finish (each workgroup takes a ticket from *gn_counter after its statistics are in;
     * the counter re-arms itself to 0).  Consumers that normalise on the fly (in_scale / in_shift of the next conv, the
     * resampling kernels) then need no launch in between.  gn_counter: one zero-initialised unsigned per image.  Paths
     * that finish the statistics in a pass of their own (split K, exact fp32) launch the table kernel internally.      */
    const float* gn_gamma; const float* gn_beta;    /* GroupNorm weight / bias, Cout floats                              */
    float* gn_scale_out; float* gn_shift_out;       /* Cout floats each; image b's tables gn_tab_bs floats behind image 0 */
    unsigned* gn_counter;
    int gn_tab_bs;
    /* ---- ABI 17, 1x1 layers on the implicit-GEMM route (f16x3) only: a layer whose epilogue carries a PREDICTED
     * normalisation of its output (otvm_gn_predict writes w_scale / bias per frame and per image).                      */
    int ws_bs;                                      /* floats between the images' w_scale / bias arrays (0 = shared)      */
    const float* res_scale;                         /* optional, with residual: residual' = residual * res_scale[c] (the
                                                       GroupNorm scale of a raw identity-path tensor, otvm_gn_table; its shift
                                                       belongs into bias -- otvm_gn_predict_params.res_shift); Cout floats,
                                                       image b's table res_scale_bs floats behind image 0                 */
    int res_scale_bs;
    const float* in_res; int in_res_ld; int64_t in_res_bs;
                                                    /* optional, with in_scale, layers for which otvm_conv2d_accepts_input_residual()
                                                       returns 1 (3x3 patch kernel, <= 64 output channels): the input is the raw
                                                       input of a residual block's last GroupNorm whose apply pass is skipped,
                                                       in' = in_act(in * in_scale[c] + in_shift[c] + in_res) -- the arithmetic of
                                                       otvm_gn_apply with a residual; in_res = the block's identity, same shape   */
} otvm_conv_params;
int otvm_conv2d_accepts_input_residual(const otvm_conv_params* p);

/* ABI 17: the last 3x3 convolution of the FBA decoder / refinement (32 -> 16, LeakyReLU; FBA/models.py:383-388, 425-432) with
 * the head that follows it -- 1x1 conv 16 -> n_out, clamp / sigmoid, fba_fusion, softmax of the trimap-refinement logits; the
 * per-pixel arithmetic of otvm_fba_head below -- in the epilogue: the 16 hidden values never leave the registers.  p = the conv
 * (3x3, stride 1, dilation 1, Cin % 16 == 0, Cout == 16, f16x3, w_frag; p->out = the hidden state or NULL: not written);
 * image b of the head's tensors lives *_bs floats behind image 0 (p->batch > 1).                                            */
typedef struct otvm_head_params {
    const float* w; const float* b; int n_out;      /* 1x1 head: [n_out][16], [n_out]; n_out = 7 or 10                         */
    const float* img; int img_ld;                   /* composited RGB in [0,1]: 3 channels at pixel stride img_ld               */
    int64_t P;                                      /* pixels per plane of the planar outputs (= H * W of the conv)            */
    float* alpha_out; int alpha_stride;             /* fused alpha, element i at alpha_out[i * alpha_stride]                   */
    float* tri_out;                                 /* n_out == 10: [3][P] softmax of logits 7..9                              */
    float* sm; int sm_ld;                           /* optional (n_out == 10): p_unknown, p_fg, alpha -> sm[i * sm_ld + 3..5]   */
    int64_t img_bs, alpha_bs, tri_bs, sm_bs;
    const void* w16;                                /* optional: the conv's weights as B fragments of the 16-wide matrix-core tile
                                                       (otvm_pack_head16_weight_f16x3; Cin == 32): v_mfma_f32_16x16x32_f16 with
                                                       N = the 16 real output channels instead of a half-empty 32-wide tile   */
} otvm_head_params;
int otvm_conv2d_head(const otvm_conv_params* p, const otvm_head_params* h, void* stream);
int64_t otvm_head16_weight_bytes_f16x3(void);
/* w_packed = otvm_pack_conv_weight's output of a 3x3 layer with I_pad == 32 and 16 filters, w_scale = its split scale */
int otvm_pack_head16_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, const float* w_scale, void* w16, void* stream);
int otvm_conv2d(const otvm_conv_params* p, void* stream);
/* The legal kernel configurations of a layer (f16x3): the patch kernel where the shape allows it, and the implicit-GEMM
 * tiles 256x256 ... 64x64, each alone or with the K range of every output tile shared by S = 2..8 workgroups
 * (deterministic fixed-order reduction through splitk_ws).  Writes up to max_n opaque codes to `out`, returns their
 * number; a code goes into otvm_conv_params.tune.  All configurations compute the same convolution (results differ
 * by fp32 summation order only); the host times them on the device once per layer shape and keeps the fastest.    */
int otvm_conv2d_candidates(const otvm_conv_params* p, int* out, int max_n);
/* 1 when otvm_conv2d would run the layer on a kernel that implements in_scale / in_shift -- f16x3: the 3x3 stride-1 patch
 * kernel, or (round 3) any implicit-GEMM tile on a layer with Cin % 32 == 0 and no in_relu -- else 0: the caller then applies
 * otvm_gn_apply as a separate pass.  Bit-identical to that two-pass route in the same configuration.                       */
int otvm_conv2d_accepts_input_norm(const otvm_conv_params* p);
/* 0 = no kernel takes it, 1 = the patch kernel would run this layer, 2 = an implicit-GEMM tile (which normalises every input
 * element once per tap: worth it for 1x1 layers, not for 3x3 layers with many channels -- the host's choice).  (ABI 16) */
int otvm_conv2d_input_norm_kind(const otvm_conv_params* p);

/* f16x3: derive the split weights from a packed fp32 weight (see otvm_pack_conv_weight):
 * row o is scaled by 2^-e (|w| <= 1), w_hi = fp16(w), w_lo = fp16(w - w_hi), w_scale[o] = 2^e.
 * When I_pad % 32 == 0 (and taps <= 32) K is re-ordered channel-block major ([c/32][tap][c%32]) so
 * that the taps of one channel block are consecutive K chunks (input re-reads hit L1/L2).        */
int otvm_split_conv_weight_f16x3(const float* w_packed, int O, int O_pad, int K_pad, int taps, int I_pad,
                                 void* w_hi, void* w_lo, float* w_scale, void* stream);

/* f16x3, layers with I_pad % 32 == 0 (K in whole 32-channel chunks): the split weights w_hi / w_lo [O_pad][K_pad] (from
 * otvm_split_conv_weight_f16x3, K already in the kernel's chunk order) re-packed as MFMA B fragments
 * [O_pad/32][K_pad/32][2 k-steps][hi|lo][64 lanes][8 halfs] (1-KiB blocks, one coalesced load each) for the one-wave
 * 64x64 tile -- the small-map configuration whose operands bypass LDS; goes into otvm_conv_params.w_wfrag.           */
int64_t otvm_wave_weight_bytes_f16x3(int O_pad, int K_pad);
int otvm_pack_wave_weight_f16x3(const void* w_hi, const void* w_lo, int O_pad, int K_pad, void* w_wfrag, void* stream);

/* One torchvision Bottleneck of the STM encoders (stride 1, eval-mode BatchNorm folded: STM.py:43-51,79-87) as ONE launch:
 * t1 = relu(W1 x + b1), t2 = relu(W2 * t1 + b2), y = relu(W3 t2 + b3 + identity); the intermediates stay in LDS, x is read once
 * (with a one-pixel halo), y written once.
 *   1/4-resolution stage (planes = 64, y has 256 channels; csrc/bottleneck_f16x3.hip):
 *     Cin = 256: identity block (identity = x).   Cin = 64: the stage's first block -- the projection Wd x is folded into the
 *     last GEMM: w3f / s3 then belong to the concatenated filter [W3 | Wd] (K = 128) and b3 = b3 + bd.
 *   1/8-resolution stage (ABI 19; planes = 128, y has 512 channels; csrc/bottleneck128_f16x3.hip):
 *     Cin = 512: identity block (res3.1 - res3.3).  `tile` picks the pixel block of a workgroup: 0 = from the map size,
 *     1 = 8 x 16, 2 = 8 x 8, 3 = 4 x 8 (every tile computes the same values; fp32 summation orders are identical too).
 *   w1f / w2f / w3f: otvm_pack_wave_weight_f16x3 of the split weights (K-major [O_pad][K_pad] from otvm_split_conv_weight_f16x3);
 *   s*: their per-filter scales; b*: folded biases.  f16x3 arithmetic (OTVM_PREC_F16X3).                                  */
typedef struct {
    const float* x; int H, W, Cin, x_ld;            /* [H*W, Cin] view                                       */
    float* y; int y_ld;                             /* [H*W, 256 | 512] view                                 */
    const void* w1f; const void* w2f; const void* w3f;
    const float* s1; const float* s2; const float* s3;
    const float* b1; const float* b2; const float* b3;
    int batch; int64_t x_bs, y_bs;                  /* images per launch, floats between consecutive images  */
    int tile;                                       /* ABI 19: planes-128 blocks only, see above             */
} otvm_stm_bottleneck_params;
int otvm_stm_bottleneck_f16x3(const otvm_stm_bottleneck_params* p, void* stream);

/* ---------------------------------------------------------------- GroupNorm(32) ----------------
 * nn.GroupNorm(32, C, eps=1e-5, affine) (layers_WS.py:26-27, FBA/models.py:272-276), two passes:
 * stats accumulates per-group sum / sum-of-squares in fp64 (stats[32][2], must be zeroed before);
 * apply computes y = act((x-mean)*rstd*gamma + beta + residual).                                  */
int otvm_gn_stats(const float* x, int64_t P, int C, int ld, double* stats, void* stream);
/* per-channel scale / shift of the same normalisation (scale = rstd*gamma, shift = beta - mean*scale), for
 * otvm_conv_params.in_scale / in_shift: identical arithmetic to otvm_gn_apply's                                   */
int otvm_gn_table(const double* stats, int64_t P, int C, const float* gamma, const float* beta, float* scale,
                  float* shift, void* stream);
/* res_scale / res_shift (optional, tables from otvm_gn_table): the residual is itself a raw GroupNorm input whose apply
 * pass was skipped; it is normalised on the fly, residual' = res_act(residual * res_scale[c] + res_shift[c]).        */
int otvm_gn_apply(const float* x, int64_t P, int C, int ld, const double* stats, const float* gamma,
                  const float* beta, const float* residual, int res_ld, const float* res_scale, const float* res_shift,
                  int res_act, int act, float* out, int out_ld, void* stream);

/* Batched forms (ABI 11): the same normalisation applied to `batch` images in one launch -- per-image statistics
 * (GroupNorm is per sample), shared gamma / beta.  Image b of x / out / residual lives *_bs floats behind image 0, its
 * [32][2] statistics block stats_bs doubles, its scale / shift tables norm_bs floats.  batch = 1 == the calls above. */
int otvm_gn_stats_b(const float* x, int64_t P, int C, int ld, double* stats, int batch, int64_t x_bs, int stats_bs, void* stream);
int otvm_gn_table_b(const double* stats, int64_t P, int C, const float* gamma, const float* beta, float* scale, float* shift,
                    int batch, int stats_bs, int norm_bs, void* stream);
typedef struct {
    const float* x; int64_t P; int C, ld; const double* stats; const float* gamma; const float* beta;
    const float* residual; int res_ld; const float* res_scale; const float* res_shift; int res_act, act;
    float* out; int out_ld;
    int batch; int64_t x_bs, res_bs, out_bs; int stats_bs, norm_bs;
} otvm_gn_apply_params;
int otvm_gn_apply_b(const otvm_gn_apply_params* p, void* stream);

/* ---- GroupNorm statistics of a 1x1 convolution's output, PREDICTED from its input (ABI 17; csrc/gram.hip) ------------
 * FBA bottleneck tail (resnet_GN_WS.py:66-86): out = relu(GroupNorm(conv3(x')) + identity), x' = relu(GroupNorm(conv2 ...)).
 * y = W x' is linear, so the sums the GroupNorm of y needs follow from the input:  sum y = v_g . s,  sum y^2 = <G, M_g>  with
 * s = sum_p x'_p, G = sum_p x'_p x'_p^T (this call) and v_g = sum_{c in g} w_c, M_g = sum_{c in g} w_c w_c^T (per checkpoint).
 * otvm_gram_f16 : x = the RAW GroupNorm input of x' ([P][ld] fp32, C channels), in_scale / in_shift / in_act = its folded
 *                 normalisation (tables of otvm_gn_table; NULL = x is x' already).  Writes per pixel chunk k (nk =
 *                 otvm_gram_chunks(P, C, &pch) chunks of pch pixels) the block-upper-triangular partial Gram matrix
 *                 gpart[k][blk][bs*bs] (bs = otvm_gram_block(C); blocks (bi <= bj) row-major; otvm_gram_entries(C) floats per
 *                 chunk) and the partial channel sums spart[k][C].  passes = 1: operands rounded to fp16 (round to nearest;
 *                 the statistics average the rounding noise out), 3: the f16x3 split of the convolutions.
 * otvm_gn_predict: adds the partials in fp64, contracts them with Mp[32][entries] (fp32, group-major; off-diagonal blocks
 *                 carrying the factor 2 of the symmetric half) and v[32][C] (fp64), and writes  scale_eff[c] = wscale[c] * rstd_g * gamma[c],
 *                 bias_eff[c] = beta[c] - mean_g * rstd_g * gamma[c] (+ res_shift[c])  -- the values the consuming conv takes
 *                 as otvm_conv_params.w_scale / bias.  ws = batch * otvm_gn_predict_ws_bytes() of workspace (per-workgroup partial
 *                 sums, added in a fixed order by the last workgroup: deterministic), counter = zeroed uint32 per image (left
 *                 zeroed); stat_out (optional) receives (mean, rstd) per group.  Batched like otvm_gn_*_b.                     */
typedef struct otvm_gram_params {
    const float* x; int64_t P; int C, ld;
    const float* in_scale; const float* in_shift; int in_act;
    float* gpart; float* spart; int passes;
    int batch; int64_t x_bs; int norm_bs;            /* image b: x + b * x_bs, tables + b * norm_bs; partials are packed per image */
    unsigned* diag;                                  /* ABI 18, optional: the layer's two diagnostic words (see otvm_gn_predict_params);
                                                        this kernel sets bit 1 of diag[1] when an operand was clamped to +-65504   */
} otvm_gram_params;
int otvm_gram_block(int C);
int64_t otvm_gram_entries(int C);
int otvm_gram_chunks(int64_t P, int C, int* pch_out);
int otvm_gram_f16(const otvm_gram_params* p, void* stream);
typedef struct otvm_gn_predict_params {
    const float* gpart; const float* spart; int64_t P; int C, Cout;
    const float* Mp; const double* v; void* ws; unsigned* counter;
    const float* wscale; const float* gamma; const float* beta; const float* res_shift;
    float* scale_eff; float* bias_eff; float* stat_out;
    int batch, tab_bs, rs_bs;                        /* image b: tables + b * tab_bs floats, res_shift + b * rs_bs floats */
    unsigned* diag;                                  /* ABI 18, optional, two words the host clears and reads: diag[0] = running maximum
                                                        (atomicMax of float bits) of kappa = mean^2 / var over groups, images and
                                                        launches -- the factor by which the Gram matrix's rounding error reaches the
                                                        variance; diag[1] bit 0 = a statistic was not finite                         */
} otvm_gn_predict_params;
int64_t otvm_gn_predict_ws_bytes(void);
int otvm_gn_predict(const otvm_gn_predict_params* p, void* stream);

/* ---------------------------------------------------------------- pooling / resampling ---------*/
/* F.max_pool2d(3, 2, 1) (resnet_GN_WS.py:98, torchvision resnet maxpool in STM.py:47,83) */
int otvm_maxpool3x3s2(const float* in, int H, int W, int C, int ld, float* out, int out_ld, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) to (Ho,Wo); out = up(in) [+ add]
 * (FBA/models.py:358-376, STM.py:115) */
/* in_scale / in_shift (optional, tables from otvm_gn_table): the GroupNorm apply of the input folded into the resampling,
 * in' = in_act(in * in_scale[c] + in_shift[c]) per source pixel (FBA/models.py:364-376: GN + LeakyReLU, then interpolate). */
int otvm_upsample_bilinear(const float* in, int Hi, int Wi, int C, int in_ld, const float* in_scale, const float* in_shift,
                           int in_act, const float* add, int add_ld, float* out, int Ho, int Wo, int out_ld, void* stream);
/* batched forms (ABI 11): `batch` images per launch, image b lives *_bs floats behind image 0 (norm_bs: the tables) */
int otvm_maxpool3x3s2_b(const float* in, int H, int W, int C, int ld, float* out, int out_ld, int batch, int64_t in_bs,
                        int64_t out_bs, void* stream);
int otvm_upsample_bilinear_b(const float* in, int Hi, int Wi, int C, int in_ld, const float* in_scale, const float* in_shift,
                             int in_act, const float* add, int add_ld, float* out, int Ho, int Wo, int out_ld, int batch,
                             int64_t in_bs, int64_t add_bs, int64_t out_bs, int norm_bs, void* stream);
/* nn.AdaptiveAvgPool2d(s) for s in {1,2,3,6} in one launch (FBA/models.py:300-306);
 * out = 50 bins x C, bins ordered scale-major then row-major.
 * ws >= otvm_ppm_pool_ws_bytes(H, C): per-row sums of the 12 column bins (one pass over the map, then a
 * fixed-order reduction over the rows of every bin).                                                */
int64_t otvm_ppm_pool_ws_bytes(int H, int C);
int otvm_ppm_pool(const float* in, int H, int W, int C, int ld, float* out, void* ws, void* stream);

/* The four PPM heads behind the pooling (FBA/models.py:298-307: L.Conv2d(2048, 256, 1, bias) -> GroupNorm(32) ->
 * LeakyReLU on each pooled map) in one launch.  pooled = otvm_ppm_pool's output [50][C]; w[i] = packed fp32 weight of
 * branch i ([>= 256 rows][K_pad], weight standardisation already applied by otvm_pack_conv_weight), bias[i] optional;
 * out[i] = [s*s][out_ld] (s = 1, 2, 3, 6), normalised and activated -- ready for otvm_upsample_bilinear.           */
typedef struct otvm_ppm_head_params {
    const float* pooled; int C, K_pad, Cout;
    const float* w[4]; const float* bias[4]; const float* gamma[4]; const float* beta[4];
    float* out[4]; int out_ld, act;
} otvm_ppm_head_params;
int otvm_ppm_head(const otvm_ppm_head_params* p, void* stream);

/* The PPM branches' share of conv_up1.0 (FBA/models.py:358-365: 3x3 conv over cat[layer4, 4 upsampled PPM maps]) WITHOUT
 * materialising the upsampled maps: convolution and bilinear interpolation are linear, so
 *   otvm_ppm_conv_z  : Z[tap][j][o] = sum_c w_ppm[scale(j)][tap][c][o] * y_scale[j][c]  for the 50 pooled pixels j (otvm_ppm_head's
 *                      outputs y[4], pixel stride y_ld, 256 channels) and the 9 taps; w_ppm fp32 [4][9][256 c][256 o]; Z [9][50][256];
 *   otvm_ppm_conv_add: out[p][o] += sum_tap [p + tap inside HxW] sum_scale bilinear_up(Z[tap][scale])(p + tap)   (256 channels).
 * Identical to the reference's conv(cat(...)) restricted to the PPM channels up to fp32 summation order.                 */
int otvm_ppm_conv_z(const float* const* y, int y_ld, const float* w_ppm, float* Z, void* stream);
int otvm_ppm_conv_add(const float* Z, int H, int W, float* out, int out_ld, double* gn_stats, void* stream);
                                  /* gn_stats (optional, zeroed [32][2] fp64): GroupNorm(32) sums of the FINAL out, as otvm_conv_params.gn_stats */

/* ---------------------------------------------------------------- memory read (STM.py:140-163) -
 * mem[q, :] = sum_m softmax_m(K[m,:].Q[q,:] / sqrt(128)) V[m,:], m over T slots x hw positions.
 * Flash-style: one partial (max, sum, acc) per (query tile, slot), merged by a combine kernel;
 * the [T*hw, hw] probability matrix of the reference is never materialised.
 *   keys[t] : [hw,128] fp32, vals[t] : [hw,512] fp32 (device pointer tables live on the HOST)
 *   out     : [hw, 512] view with pixel stride out_ld (the first half of the 1024-ch m4 tensor)
 *   ws      : workspace >= otvm_memory_read_ws_bytes(hw, T) bytes                                 */
int64_t otvm_memory_read_ws_bytes(int hw, int T);
int otvm_memory_read(const float* q_key, int q_ld, const float* const* keys, const float* const* vals, int T,
                     int hw, float* out, int out_ld, void* ws, void* stream);

/* f16x3 variant (fp32-class accuracy on the f16 MFMA, see OTVM_PREC_F16X3).  The bank is this build's own
 * structure, so each slot is stored split (fp16 hi/lo) and in MFMA fragment order: otvm_bank_pack_f16x3
 * converts the fp32 key [hw,128] / value [hw,512] maps of a memorised frame (the KV_M_r4 conv outputs,
 * STM.py:201-228) into one packed slot of otvm_bank_slot_bytes_f16x3(hw) bytes; the read kernel streams
 * slots with coalesced 1-KiB wave loads directly into MFMA operands.  ws as otvm_memory_read.            */
int64_t otvm_bank_slot_bytes_f16x3(int hw);
int otvm_bank_pack_f16x3(const float* key, const float* val, int hw, void* slot, void* stream);
int otvm_memory_read_f16x3(const float* q_key, int q_ld, const void* const* slots, int T, int hw, float* out,
                           int out_ld, void* ws, void* stream);

/* The same read in two steps, for a caller that knows part of the bank earlier than the rest (the engine reads the slots
 * that are already resident on a side stream, under the previous frame's alpha network, and only the slot of the frame
 * just memorised on the critical path): the softmax over the memory axis is merged from per-chunk partials (running
 * max, sum, weighted value sum), so the bank may be visited in any grouping.  A workspace laid out for np_cap partials
 * holds np_cap * hw * (512 + 2) floats; a group of n slots writes otvm_memory_read_f16x3_partial_count(n, hw) partials
 * starting at part0 (*part_end = one past its last); combine merges partials [0, n_partials) into out.             */
int otvm_memory_read_f16x3_partial_count(int n_slots, int hw);
int otvm_memory_read_f16x3_partial(const float* q_key, int q_ld, const void* const* slots, int n_slots, int hw, void* ws,
                                   int np_cap, int part0, int* part_end, void* stream);
int otvm_memory_read_f16x3_combine(const void* ws, int np_cap, int n_partials, int hw, float* out, int out_ld, void* stream);

/* ---------------------------------------------------------------- frame glue --------------------
 * preprocess: alpha/model.py:380-389,408-414 + STM.py:53-57,89-93.  fg,bg: [3,H,W] fp32 BGR 0..255
 * planes; a: [H,W] in [0,1].  Writes the zero-padded 0..1 RGB composite and its normalised copies
 * into channel slices of the consumers' input buffers and the un-padded `scaled_imgs` [3,H,W].
 * Every destination (scaled_imgs, x11, sq, sm, d80) is optional: NULL = not written by this call.  */
typedef struct {
    const float* fg; const float* bg; const float* a;
    int H, W, Hp, Wp, lh, lw;
    float mean[3], std[3];        /* IMG_MEAN / IMG_STD                    */
    float mean_q[3], std_q[3];    /* trimap.model.Encoder_Q.mean / std     */
    float mean_m[3], std_m[3];    /* trimap.model.Encoder_M.mean / std     */
    float* scaled_imgs;           /* [3,H,W] planar RGB 0..1 (returned)    */
    float* x11; int x11_ld;       /* ch0-2 <- normalised                   */
    float* sq;  int sq_ld;        /* ch0-2 <- Encoder_Q normalised         */
    float* sm;  int sm_ld;        /* ch0-2 <- Encoder_M normalised         */
    float* d80; int d80_ld;       /* ch64-66 <- normalised, ch67-69 <- 0..1 */
    const unsigned char* fg_u8;   /* optional: decoded frames as uint8 [H,W,3] (interleaved, what an image decoder  */
    const unsigned char* bg_u8;   /* hands out); when set, fg / bg are ignored.  float(uint8) == the reference's     */
    int u8_rgb;                   /* .float() of the same pixels; u8_rgb != 0: channels are R,G,B instead of B,G,R   */
} otvm_preprocess_params;
int otvm_preprocess(const otvm_preprocess_params* p, void* stream);

/* pad a [3,H,W] one-hot/soft trimap to [3,Hp,Wp] (bg plane padded with 1, others 0;
 * alpha/model.py:410) */
int otvm_pad_trimap(const float* tri, int H, int W, float* out, int Hp, int Wp, int lh, int lw, void* stream);

/* STM decoder tail: x4 bilinear upsample of the [h4*w4,3] logits (ld) + softmax over the 3 classes
 * -> planar probs [3,Hp,Wp] (STM.py:136, alpha/model.py:440) */
int otvm_upsample4_softmax3(const float* logits, int h4, int w4, int ld, float* probs, void* stream);

/* 8-channel trimap encoding (alpha/model.py:40-53, utils/utils.py:12-39): argmax class map, exact
 * Euclidean distance transform of the bg / fg classes on device (integer d^2), three Gaussians each,
 * plus the two soft channels.  probs: planar [3,Hp,Wp].  Writes x11 ch3-10 and d80 ch70-71.
 *   cls_override: optional u8 class map [Hp*Wp] used INSTEAD of the argmax (tests only), may be NULL
 *   cls_out     : u8 class map [Hp*Wp] (always written)
 *   ws          : workspace >= otvm_trimap_encode_ws_bytes(Hp,Wp)                                  */
int64_t otvm_trimap_encode_ws_bytes(int Hp, int Wp);
int otvm_trimap_encode(const float* probs, int Hp, int Wp, const uint8_t* cls_override, uint8_t* cls_out,
                       float* x11, int x11_ld, float* d80, int d80_ld, void* ws, void* stream);

/* FBA / refinement heads: 1x1 conv 16->7 (+3 trimap logits when n_out==10), clamp/sigmoid and
 * fba_fusion (FBA/models.py:279-288,383-388,425-432), softmax of the trimap logits
 * (alpha/model.py:460).  hid: [P,16] view (ld).  img: [P,3] view (0..1 RGB).
 *   alpha_out/alpha_stride : fused alpha written at alpha_out[p*alpha_stride]
 *   tri_out  : planar [3,P] softmax probs (n_out==10) or NULL
 *   sm       : Encoder_M input buffer; ch3 <- p_un, ch4 <- p_fg, ch5 <- alpha (n_out==10) or NULL */
int otvm_fba_head(const float* hid, int hid_ld, const float* w, const float* b, int n_out, const float* img,
                  int img_ld, int64_t P, float* alpha_out, int alpha_stride, float* tri_out, float* sm, int sm_ld,
                  void* stream);

/* crop the padding and produce the returned tensors (alpha/model.py:495-508, eval.py:209):
 * alpha [H,W] fp32, alpha_u8 [H,W] = trunc(alpha*255) (may be NULL), trimap [3,H,W] (may be NULL) */
int otvm_crop_outputs(const float* alpha_p, const float* tri_p, int Hp, int Wp, int H, int W, int lh, int lw,
                      float* alpha, uint8_t* alpha_u8, float* tri, void* stream);

/* first-frame trimap from a GT alpha (alpha/model.py:342-362): unknown = dilate(0<a<1) with a
 * (2r+1)^2 max filter, fg = (a==1), bg = (a==0); out planar one-hot [3,H,W]; ws >= H*W bytes      */
int otvm_trimap_from_alpha(const float* a, int H, int W, int r, float* out, void* ws, void* stream);

/* one-hot of the argmax of a planar [3,H,W] trimap (alpha/model.py:356-362, returned tri_gt) */
int otvm_onehot_argmax3(const float* tri, int64_t P, float* out, void* stream);

/* ---------------------------------------------------------------- metrics ------
 * SAD / MSE / dtSSD partial sums of one frame (reference utils/tmp/metric.py:177-189,252-264) on 8-bit alphas
 * [n] = trunc(alpha*255) (eval.py:209); mask (unknown band) and the prev_* pointers may be NULL.  acc[5] (fp64,
 * caller-zeroed) accumulates  sum|p-t|m, sum(p-t)^2 m, sum m, sum((p-p')-(t-t'))^2 m', sum m'  -- integer-exact. */
int otvm_matting_metrics(const uint8_t* pred, const uint8_t* target, const uint8_t* mask, const uint8_t* prev_pred,
                         const uint8_t* prev_target, const uint8_t* prev_mask, int64_t n, double* acc, void* stream);

/* ---------------------------------------------------------------- range guard / clear -----------
 * f16x3 splits fp32 operands into fp16 halves: |x| >= 65504 loses accuracy, >= 131008 becomes inf.
 * otvm_finite_guard scans an NHWC view [P, C] (ld) and stores min(*flag, tag) when any element fails |x| < limit
 * (NaN fails too); *flag is initialised to INT32_MAX by the caller and read by the host when it synchronises.  The
 * engine runs it on everything that survives a frame: the memorised key / value maps (STM.py:201-228), the hidden
 * state and the propagated trimap logits (alpha/model.py:432-471).                                               */
int otvm_finite_guard(const float* x, int64_t P, int C, int ld, float limit, int tag, int* flag, void* stream);
/* stream-ordered zero fill (hipMemsetAsync) -- the GroupNorm statistics arena is cleared once per frame           */
int otvm_clear(void* p, int64_t bytes, void* stream);

/* ---------------------------------------------------------------- training forward -------------
 * FullModel.forward of the reference's training class (models/alpha/model.py:189-312), FORWARD ONLY: the network runs on
 * the kernels above (B clips in lock-step, memory every frame); these entry points add what the losses need.
 * All tensors planar fp32; "N images" = batch x frames (x channels where noted); sums leave in fp64 accumulators that
 * the caller zeroes and divides by the element counts torch.mean / mse_loss / CrossEntropyLoss use.               */
/* the heads again with all outputs: out7 planar [7][P] = fused alpha, F, B (FBA/models.py:383-388,425-432);
 * logits_out planar [3][P] = the refinement's trimap logits (n_out == 10)                                           */
int otvm_fba_head_train(const float* hid, int hid_ld, const float* w, const float* b, int n_out, const float* img, int img_ld,
                        int64_t P, float* out7, float* logits_out, void* stream);
/* the STM decoder's logits, x4 bilinear (STM.py:136), WITHOUT the softmax: planar [3][Hp*Wp]                         */
int otvm_upsample4_logits3(const float* logits, int h4, int w4, int ld, float* logits_out, void* stream);
/* frame 0 of a training clip memorises the GROUND-TRUTH trimap (model.py:212, preds_trimap_refine[0] = tri[:,0]):
 * planar [3][P] -> the unknown / foreground channels of the Encoder_M input buffer                                   */
int otvm_trimap_to_sm(const float* tri, int64_t P, float* sm, int sm_ld, void* stream);
/* model.py:59-60: out = in.flip(channel) * s on planar [N][3][P]; model.py:42-44: trimask = (argmax == unknown), class map */
int otvm_scale_flip3(const float* in, int64_t N, int64_t P, float s, float* out, void* stream);
int otvm_trimask(const float* tri, int64_t N, int64_t P, float* mask, unsigned char* cls, const float* gts, float* vis, void* stream);
                                                    /* vis (optional): where(trimask, 128/255, gts), model.py:296-300 */
/* fba_single_image_loss, per-pixel part (model.py:117-150): writes cF, cB, comp [N][3][P], alpha_out [N][P]; acc5 += sums of |a - gt|,
 * |cF gt + cB (1-gt) - img|, |fgs a + bgs (1-a) - img|, |cF - fgs|, |cB - bgs|                                      */
int otvm_loss_fba_comp(const float* pred7, const float* gt, const float* trimask, const float* fgs, const float* bgs, const float* img,
                       int64_t N, int64_t P, float* cF, float* cB, float* comp, float* alpha_out, double* acc5, void* stream);
/* L1_grad (utils/loss_func.py:44-51): acc += | |grad x| - |grad y| | over planar [N][H][W]                            */
int otvm_loss_grad_l1(const float* x, const float* y, int64_t N, int H, int W, float eps, double* acc, void* stream);
/* exclusion_loss (loss_func.py:56-82), one pyramid level over images [B][S][3][H][W]: acc1[S][4] (zeroed) receives the
 * per-frame sums of |gx1|, |gy1|, |gx2|, |gy2|, acc2[B*S][2] (zeroed) the per-sample sums of the squared products       */
int otvm_loss_exclusion_level(const float* img1, const float* img2, int B, int S, int H, int W, float eps, double* acc1, double* acc2,
                              void* stream);
int otvm_avgpool2(const float* x, int64_t N, int H, int W, float* y, void* stream);
/* LapLoss (loss_func.py:95-155), one level: down = gauss(cur)[::2, ::2] for image and target (written), acc += weight *
 * | (cur_i - up(down_i)) - (cur_t - up(down_t)) |                                                                    */
int otvm_loss_lap_level(const float* cur_img, const float* cur_tgt, int64_t N, int H, int W, double weight, float* down_img,
                        float* down_tgt, double* acc, void* stream);
/* model.py:177-182: acc += ((x[b,t+1]-x[b,t]) - (y[b,t+1]-y[b,t]))^2 over planar [B][S][CP]                           */
int otvm_loss_temporal(const float* x, const float* y, int B, int S, int64_t CP, double* acc, void* stream);
/* nn.CrossEntropyLoss (model.py:286-290): acc += -log_softmax(logits)[cls] over planar logits [N][3][P], cls [N][P] u8 */
int otvm_loss_ce3(const float* logits, const unsigned char* cls, int64_t N, int64_t P, double* acc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTVM_HIP_H */
