/* rcf_hip.h -- C ABI of librcf_hip.so: the MI355X (gfx950) kernels of RCF's training hot path.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types.  `stream` is a hipStream_t passed as void*.
 *   - every entry point returns 0 on success, a positive hipError_t on a HIP failure, or a
 *     negative RCF_E* code on a bad argument.  Nothing calls exit() (the reference's
 *     tools/torchCRF/src/permutohedral_gpu.cu:44-53 kills the process on a CUDA error).
 *   - activations are NHWC fp32 ("pixel-major": element (n,y,x,c) at ((n*H+y)*W+x)*pitch + c,
 *     pitch >= C so a tensor may be a channel slice of a wider buffer); conv weights are
 *     [Cout][R][S][Cin] (= a torch OIHW tensor in channels_last memory format).
 *   - nothing allocates: scratch is passed in, sized by the matching *_workspace_bytes().
 *
 * Each group cites the reference interface it stands in for (paths under the reference repo).
 */
#ifndef RCF_HIP_H
#define RCF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCF_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define RCF_EWORKSPACE (-2) /* workspace too small */

/* storage type codes of the mixed-precision (`*_mp`, `*_bf16`) entry points: activations and activation gradients may
 * be held as bf16 between layers (BASELINE configs[2]: bf16 forward / fp32 gradients); every kernel computes in fp32.
 * Pitches are in ELEMENTS of the tensor's own type. */
#define RCF_F32 0
#define RCF_BF16 1

/* library identity: "rcf_hip <version> gfx950" */
const char *rcf_version(void);

/* ---- convolution as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) ------------------------
 * Replaces torch.nn.Conv2d / cuDNN for every conv on the path: models/resnet.py:164-203,565-572,
 * models/res_layer.py:54-60, mmcv ConvModule in models/fcn_head.py:107-130, conv_seg :100-101,
 * flow_feat_before_agg models/flow_aggregation_head_with_residual.py:84-93.
 * Requirements: Cin % 4 == 0 (3-channel / 2-channel inputs are zero-padded to 4 by the caller),
 * pitches % 4 == 0, pointers 16-byte aligned.  Square kernels/strides/dilations/pads (all the
 * path uses).  act: 0 none, 1 LeakyReLU(slope).  bias may be NULL.  beta: 0 overwrite, 1 accumulate. */
typedef struct {
    int N, H, W, Cin;        /* input  [N,H,W,Cin], pixel pitch x_pitch */
    int Ho, Wo, Cout;        /* output [N,Ho,Wo,Cout], pixel pitch y_pitch */
    int R, S, stride, pad, dil;
    int x_pitch, y_pitch;
    /* Optional operand ranges: device scalars holding max |value| of x / w / dy as raw fp32 bits (rcf_absmax_f32);
     * NULL = unknown.  When the two a launch contracts are known (fwd: x, w; dgrad: dy, w; wgrad: x, dy) the
     * fp16-pair kernels run -- operands scaled by powers of two into fp16's range, split into 2 x fp16, 3 partial
     * products on the matrix cores (fp32-level error, see csrc/igemm_conv.hip) -- otherwise the bf16-triple kernels
     * (6 partial products).  A stale range that is too SMALL for its tensor overflows fp16: ranges must be fresh. */
    const unsigned *amax_x, *amax_w, *amax_dy;
    /* Optional, forward launches of the fp16-pair kernels: w already split by rcf_conv_weight_pairs_f32 with the
     * range amax_w points to (the kernel then reads the two fp16 planes instead of splitting w in every row tile) */
    const void *w_pairs;
    /* Optional, data-gradient launches: the transposed weight operand prepared ONCE per weight update instead of per
     * launch -- fp32 entry points: rcf_conv_weight_pairs_t_f32 (fp16 planes of the transposed weights, with amax_w);
     * bf16 entry points: rcf_conv_weight_bf16(..., transpose = 1).  The workspace of the launch may then be NULL. */
    const void *w_pairs_t;
    /* Optional, preferred over w_pairs / w_pairs_t: rcf_conv_weight_pairs2_f32 buffers (transpose 0 / 1) holding the split
     * weights in BOTH reading orders; the wide, deep convs then run on the persistent LDS-DMA kernel (conv_h2p_kernel,
     * csrc/igemm_h2p.inc), the others on the kernels that read w_pairs / w_pairs_t.  Results are identical. */
    const void *w_pairs2, *w_pairs2_t;
    /* Optional: a zeroed device scalar that receives (atomicMax) the raw bits of max |value| of the tensor the launch WRITES
     * (forward: y; data gradient: dx, after the accumulation when beta = 1) -- the next consumer's operand range without a pass
     * over the tensor.  Exact only for a launch that writes the whole tensor (no region). */
    unsigned *amax_y;
    /* RCF_CONV_* bits below; 0 = the library's own choices.  Per call, not process state: two host threads may launch with
     * different flags at the same time. */
    unsigned flags;
    /* sizeof(rcf_conv_shape) as the CALLER compiled it.  Every entry point that takes the struct refuses (RCF_EINVAL) a size
     * other than its own: a caller built against an older, shorter header cannot make the library read past its struct. */
    unsigned struct_bytes;
} rcf_conv_shape;

/* weight gradient on 128 x 256 tiles (two workgroups per CU) even where the 256 x 256 tile (one workgroup per CU, 1.02 - 1.08 x
 * alone on the chip) applies: for a launch that shares the chip with an HBM-bound kernel on another stream, whose workgroups
 * need slots on the CUs (the late weight-gradient schedule of the training step) */
#define RCF_CONV_WGRAD_TILE_128 0x1u
/* Pre-split activation operands ("pair planes").  A tensor [N,H,W,C] (C % 16 == 0, contiguous) is stored as
 * [pixel][h: C fp16 | m: C fp16] -- x * 2^k = h + m, the byte size and pixel pitch of the fp32 tensor it replaces -- with
 * k = 14 - floor(log2(bound)) from the range scalar of the operand (amax_x / amax_dy), which for such a tensor is an UPPER BOUND
 * its producer knew before writing it (rcf_bn_apply_mp / rcf_bn_bwd_apply_mp with RCF_BN_*_PLANES emit them).  The conv kernels
 * then take both operands by LDS-DMA with no split of their own (csrc/igemm_h2d.inc; weights: rcf_conv_weight_pairs2_f32).
 *   X_PLANES:  `x` of a forward / weight-gradient launch is in that format;   DY_PLANES: `dy` of a data- / weight-gradient launch.
 * Needs the matching operand ranges and w_pairs2 / w_pairs2_t; RCF_EINVAL for shapes the plane kernels do not take. */
#define RCF_CONV_X_PLANES 0x2u
#define RCF_CONV_DY_PLANES 0x4u
/* A/B and test switches (results are bit-identical, or identical up to the re-association of fp32 sums, in every setting):
 *   H2P_NEVER / H2P_ALWAYS   the persistent LDS-DMA kernel (conv_h2p_kernel) never / whenever the shape is eligible (built-in
 *                            rule: the deep 3x3 layers: K >= 2304, >= 32768 rows, <= 2 column tiles per 2304 of K)
 *   NO_WGRAD_XCD             weight-gradient workgroups in plain grid order instead of XCD-aware (an XCD's workgroups share
 *                            their pixel range)
 *   NO_COLMAP                forward / data-gradient grids always as row bands per XCD (default: convs whose weight operand
 *                            exceeds the XCDs' L2 many times over give every XCD its own column tiles)
 *   KORDER_NATURAL           K runs tap-outer (the weight's memory order) instead of (channel chunk, tap, channel in chunk) on
 *                            the 3x3 layers with >= 256 channels per tap.  The derived weight operands are written in the
 *                            order the kernels walk: pass the same flag to the rcf_conv_weight_* / rcf_conv_weights_prepare_*
 *                            call that builds them.
 *   FP32_MFMA(v)             the exact fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32) with tuning variant v = 0..3 (bit 0:
 *                            K-step 32, bit 1: row-major LDS tiles) instead of the fp16-pair / bf16-triple kernels; no regions,
 *                            no fused statistics on this path */
#define RCF_CONV_H2P_NEVER 0x8u
#define RCF_CONV_H2P_ALWAYS 0x10u
#define RCF_CONV_NO_WGRAD_XCD 0x20u
#define RCF_CONV_NO_COLMAP 0x40u
#define RCF_CONV_KORDER_NATURAL 0x80u
/* 1x1 convs with 4 / 8 / 16 output channels on 64 .. 512 input channels (the decode heads' classifiers) run as streaming fp32
 * passes (csrc/thin.hip) instead of GEMM tiles with a handful of live columns; this bit keeps them on the GEMM kernels (A/B, tests) */
#define RCF_CONV_NO_THIN 0x100u
/* the forward / data-gradient tile kernels write their output with non-temporal (streaming) stores: the tile's 64 KB of output
 * does not stay in the XCD's 4 MB L2, where -- with 64 tiles in flight per XCD -- it evicts the weights and the activation rows
 * the neighbouring column tiles are still reading (tools/lab/gemm_lab, profiles/r06_gemm_lab_nt_stores.txt).  Results are
 * bit-identical; a per-call choice (rcf_amd.layers decides per layer: SCHED.nt_stores). */
#define RCF_CONV_NT_STORES 0x200u
#define RCF_CONV_FP32_MFMA(v) ((((unsigned)(v) & 3u) + 1u) << 12)

/* planes (rcf_conv_weight_pairs_bytes): fp16 h and m of w * 2^k, k from *amax_w, in the kernel's reading order
 * [K/16][Cout][4][h0 h1 h2 h3 m0 m1 m2 m3] with K = R*S*Cin padded to a multiple of 16 */
size_t rcf_conv_weight_pairs_bytes(int Cout, int Cin, int R, int S);
int rcf_conv_weight_pairs_f32(const float *w, int Cout, int Cin, int R, int S, const unsigned *amax_w, void *planes,
                              unsigned flags, void *stream);

size_t rcf_conv_weight_pairs2_bytes(int Cout, int Cin, int R, int S, int transpose);
int rcf_conv_weight_pairs2_f32(const float *w, int Cout, int Cin, int R, int S, int transpose, const unsigned *amax_w,
                               void *planes, unsigned flags, void *stream);
/* Batched weight preparation: the derived operands of EVERY conv weight of a model in three (fp16 pairs: ranges, forward
 * buffers, transposed buffers) or two (bf16: forward, transposed) launches instead of one per weight and layout -- they are
 * rebuilt after every optimizer step.  tab_*: device arrays of n entries {const float *w; void *out; unsigned *amax; int Cout,
 * Cin, RS, first_block, nblocks, flags; int pad[2]} (56 bytes; csrc/rcf_common.h rcf_wprep_entry) -- block b of a launch
 * serves the entry with first_block <= b < first_block + nblocks; flags bit 0 = also write the plane-separated half
 * (rcf_conv_pairs2_useful).  Outputs are byte-identical to rcf_absmax_f32 / rcf_conv_weight_pairs2_f32 / rcf_conv_weight_bf16. */
int rcf_conv_pairs2_useful(int Cout, int Cin, int R, int S, int transpose);
int rcf_conv_weights_prepare_f32(const void *tab_absmax, int blocks_absmax, const void *tab_pairs, int blocks_pairs,
                                 const void *tab_pairs_t, int blocks_pairs_t, int n, unsigned *amax_base, unsigned flags,
                                 void *stream);
int rcf_conv_weights_prepare_bf16(const void *tab_fwd, int blocks_fwd, const void *tab_t, int blocks_t, int n, unsigned flags,
                                  void *stream);

/* amax[0] = max(amax[0], bits(max |x|)) over [rows][C] (row pitch `pitch`); the caller zeroes amax[0] first */
int rcf_absmax_f32(const float *x, long rows, int C, int pitch, unsigned *amax, void *stream);

int rcf_conv2d_fwd_f32(const float *x, const float *w, const float *bias, float *y, const rcf_conv_shape *s,
                       int act, float slope, int beta, void *stream);
/* The same operators restricted to a rectangle of pixels (all channels): `region` is in OUTPUT coordinates for
 * fwd (only those pixels of y are written) and wgrad (only those pixels of dy/x-patches contribute), in INPUT
 * coordinates for dgrad (only those pixels of dx are written).  NULL = the whole tensor.  Used to evaluate
 * conv(bilinear_2x(x)) as bilinear_2x(conv_half_dilation(x)) in the interior and directly on the border band
 * (models/fcn_head.py:211-218 with input_transform='resize_concat'). */
typedef struct {
    int y0, x0, h, w;     /* rectangle */
    int band;             /* 0: the whole rectangle; t > 0: only its border frame of thickness t (2t < h, w) */
} rcf_conv_region;
/* Which kernel family a forward (dgrad = 0) / data-gradient (dgrad = 1) launch with this shape, these operand pointers and
 * flags takes: 0 the fp32-MFMA kernels, 1 the 128 x 256-tile family, 2 conv_h2p_kernel.  A pure function of its arguments --
 * nothing is launched, no state is read (profiling labels). */
int rcf_conv_kernel_of(const rcf_conv_shape *s, const rcf_conv_region *region, int dgrad);
int rcf_conv2d_fwd_region_f32(const float *x, const float *w, const float *bias, float *y, const rcf_conv_shape *s,
                              const rcf_conv_region *region, int act, float slope, int beta, void *stream);
/* y = conv(x, w) (no bias, whole tensor) and, from the same kernel's epilogue, the batch-norm statistics of y:
 * sums[2*Cout] (doubles, [sum | sumsq], the layout of rcf_bn_stats_f32; a lane's <= 64 values of a column are added
 * in fp32, everything above that in fp64) -- the conv -> (Sync)BatchNorm pairs of
 * models/resnet.py:268-300 and mmcv ConvModule without the statistics pass re-reading y.  Split-bf16 kernels only
 * (RCF_EINVAL otherwise: the caller then runs rcf_bn_stats_f32). */
size_t rcf_conv2d_fwd_stats_workspace_bytes(const rcf_conv_shape *s);
/* the transposed ([Cin][R][S][Cout], K padded to whole K-steps) fp16-pair planes the data gradient contracts against;
 * rcf_conv2d_dgrad_workspace_bytes(s) bytes */
int rcf_conv_weight_pairs_t_f32(const float *w, int Cout, int Cin, int R, int S, const unsigned *amax_w, void *planes,
                                unsigned flags, void *stream);
int rcf_conv2d_fwd_stats_f32(const float *x, const float *w, float *y, const rcf_conv_shape *s, double *sums,
                             void *workspace, size_t workspace_bytes, void *stream);
/* What a training-mode batch norm derives from its statistics (the per-channel constants of the normalisation, the
 * running statistics with momentum, torch's num_batches_tracked): handed to a statistics-producing entry point, the
 * final reduction kernel writes these too and no separate finalize launch is needed.  All pointers are device memory;
 * running_mean / running_var / num_batches_tracked may be NULL. */
typedef struct rcf_bn_finalize {
    double count;                 /* rows the statistics cover (N*H*W) */
    float eps, momentum;
    float *mean, *invstd;         /* out [C] */
    float *running_mean, *running_var;
    long long *num_batches_tracked;
} rcf_bn_finalize;
/* rcf_conv2d_fwd_stats_f32 / the statistics form of rcf_conv2d_fwd_bf16 with the batch norm finalized by the same
 * reduction launch (fin != NULL; sums may then be NULL).  fin == NULL: as the plain entry points. */
int rcf_conv2d_fwd_bnstats_f32(const float *x, const float *w, float *y, const rcf_conv_shape *s, double *sums,
                               const rcf_bn_finalize *fin, void *workspace, size_t workspace_bytes, void *stream);
int rcf_conv2d_fwd_bnstats_bf16(const void *x, const void *w_bf16, void *y, int ydt, const rcf_conv_shape *s, double *sums,
                                const rcf_bn_finalize *fin, void *workspace, size_t workspace_bytes, void *stream);
/* out[i] = sum_k partial[k][i], k < chunks, i < n, in a fixed order; scratch (64 * n doubles, may be NULL) lets a
 * long list of partial rows be summed in two levels */
int rcf_sum_partials_f64(const double *partial, int chunks, int n, double *out, double *scratch, void *stream);
/* dx[N,H,W,Cin] (pitch x_pitch) (+)= conv_transpose(dy[N,Ho,Wo,Cout] (pitch y_pitch), w).
 * `workspace` (rcf_conv2d_dgrad_workspace_bytes) holds the transposed weights [Cin][R][S][Cout] the
 * split-bf16 kernels contract against. */
size_t rcf_conv2d_dgrad_workspace_bytes(const rcf_conv_shape *s);
int rcf_conv2d_dgrad_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta,
                         void *workspace, size_t workspace_bytes, void *stream);
int rcf_conv2d_dgrad_region_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s,
                                const rcf_conv_region *region, int beta, void *workspace, size_t workspace_bytes,
                                void *stream);
/* The data gradient of a conv whose INPUT is the output of a training-mode batch norm + ReLU (the bottlenecks' bn1 -> conv2,
 * bn2 -> conv3, bn3 + identity -> the next block's conv1; models/resnet.py:268-302), and, from the same kernel's epilogue, the
 * two sums that norm's backward needs: sums2[2*Cin] = [sum g | sum g * xhat] with g = dx masked by the norm's ReLU sign bits
 * (dx after the accumulation when beta = 1: the call must be the LAST writer of dx) and xhat = (x - mean) * invstd of the norm's
 * input x -- what rcf_bn_bwd_reduce_mp would read dx and x back for.  A lane's <= 64 values of a column are added in fp32,
 * everything above in fp64, in a fixed order.  rcf_conv2d_dgrad_bnsums_ok(s): 1 when the launch this shape takes has that epilogue
 * (fp16-pair kernels with prepared weights -- w_pairs_t / w_pairs2_t and both ranges --, Cin = 64, 128 or a multiple of 256);
 * otherwise the entry point returns RCF_EINVAL before launching anything and the caller runs the two passes. */
typedef struct rcf_bn_bwd_in {
    const float *x;                 /* the norm's input [N*H*W][x_pitch] */
    int x_pitch;
    const unsigned char *relu_mask; /* [N*H*W][Cin/4] as rcf_bn_apply_mp writes it */
    const float *mean, *invstd;     /* [Cin] */
} rcf_bn_bwd_in;
int rcf_conv2d_dgrad_bnsums_ok(const rcf_conv_shape *s);
size_t rcf_conv2d_dgrad_bnsums_workspace_bytes(const rcf_conv_shape *s);
int rcf_conv2d_dgrad_bnsums_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta,
                                const rcf_bn_bwd_in *bn, double *sums2, void *workspace, size_t workspace_bytes, void *stream);
/* The same launch with a masked addend: dx = conv_transpose(dy, w) + (add_mask ? add : 0), add [N*H*W][add_pitch], add_mask
 * [N*H*W][Cin/4] (rcf_bn_apply_mp's sign bits) -- the conv1 of a bottleneck whose input also feeds the block's identity branch
 * (models/resnet.py:268-302): the identity's gradient IS the join's output gradient under the join's ReLU mask, so the batch norm's
 * backward need not write it (rcf_bn_bwd_apply_mp with dres = NULL: one tensor write less per identity block) and this launch need
 * not read it back -- it reads the join's output gradient instead.  Bit-identical to dres + accumulate.  beta must be 0 with an
 * addend; bn / sums2 / workspace as in rcf_conv2d_dgrad_bnsums_f32 or all NULL; same shapes (rcf_conv2d_dgrad_bnsums_ok). */
int rcf_conv2d_dgrad_add_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta, const float *add,
                             int add_pitch, const unsigned char *add_mask, const rcf_bn_bwd_in *bn, double *sums2,
                             void *workspace, size_t workspace_bytes, void *stream);
/* dw[Cout][R][S][Cin] (+)= sum_pixels dy * x.  Split over pixels into `workspace`, then reduced
 * deterministically (no float atomics). */
size_t rcf_conv2d_wgrad_workspace_bytes(const rcf_conv_shape *s);
int rcf_conv2d_wgrad_f32(const float *x, const float *dy, float *dw, const rcf_conv_shape *s, int beta,
                         void *workspace, size_t workspace_bytes, void *stream);
size_t rcf_conv2d_wgrad_region_workspace_bytes(const rcf_conv_shape *s, const rcf_conv_region *region);
int rcf_conv2d_wgrad_region_f32(const float *x, const float *dy, float *dw, const rcf_conv_shape *s,
                                const rcf_conv_region *region, int beta, void *workspace, size_t workspace_bytes,
                                void *stream);

/* ---- the same three operators with bf16 OPERANDS and fp32 accumulation (csrc/igemm_bf16.hip) -----------------------
 * The mixed-precision training step (BASELINE configs[2]; the reference's AMP configs configs/rcf_stv2/rcf_stage1.yaml:57-60):
 * x, dy, dx are NHWC bf16 (channel counts and pitches multiples of 8, pitches in bf16 elements), weights stay fp32
 * master copies ([Cout][R][S][Cin]) and are cast per launch, dw is fp32.  One MFMA pass (v_mfma_f32_32x32x16_bf16).
 * rcf_conv_shape's amax / w_pairs members are ignored.
 * rcf_conv_weight_bf16: the kernels' weight operand, K-step major [K/32][rows][32] bf16 (K zero-padded to a multiple of
 * 32); transpose = 0: rows = Cout, k = (r, s, cin) -- forward; transpose = 1: rows = Cin, k = (r, s, cout) -- data gradient. */
size_t rcf_conv_weight_bf16_bytes(int Cout, int Cin, int R, int S, int transpose);
int rcf_conv_weight_bf16(const float *w, int Cout, int Cin, int R, int S, int transpose, void *out, unsigned flags,
                         void *stream);
/* y (ydt: RCF_BF16 or RCF_F32, pitch s->y_pitch in its own elements) (+)= conv(x, w) + bias, act 0 none / 1 LeakyReLU.
 * w_bf16 = rcf_conv_weight_bf16(w, ..., transpose 0).  region: as rcf_conv2d_fwd_region_f32 (NULL = whole tensor).
 * sums != NULL (needs region == NULL, bias == NULL, act == beta == 0 and the workspace): the batch-norm statistics of the
 * fp32 accumulators [sum | sumsq] per output channel from the epilogue, like rcf_conv2d_fwd_stats_f32. */
size_t rcf_conv2d_fwd_stats_bf16_workspace_bytes(const rcf_conv_shape *s);
int rcf_conv2d_fwd_bf16(const void *x, const void *w_bf16, const float *bias, void *y, int ydt, const rcf_conv_shape *s,
                        const rcf_conv_region *region, int act, float slope, int beta, double *sums, void *workspace,
                        size_t workspace_bytes, void *stream);
/* dx (bf16) (+)= conv_transpose(dy (bf16), w (fp32 master)); the workspace receives the transposed bf16 weights */
size_t rcf_conv2d_dgrad_bf16_workspace_bytes(const rcf_conv_shape *s);
int rcf_conv2d_dgrad_bf16(const void *dy, const float *w, void *dx, const rcf_conv_shape *s, const rcf_conv_region *region,
                          int beta, void *workspace, size_t workspace_bytes, void *stream);
/* ---- 1x1 conv -> training-mode batch norm -> (+ residual) -> ReLU as ONE tile (csrc/foldbn.hip, csrc/igemm_bf16.hip) ----------
 * Replaces, in the bf16 step, the bottleneck tail `out = conv3(out); out = norm3(out); out += identity; out = relu(out)`
 * (models/resnet.py:281-296) and the 1x1 downsample conv -> norm (models/res_layer.py:53-63) WITHOUT the conv output or its
 * gradient ever existing in memory.  A 1x1 conv is linear in its input, so the norm's statistics and both of its backward sums
 * are functions of two small moments of the conv's INPUT x [rows][K]: the column sums A1 [K] (rcf_colsum_mp) and the Gram
 * matrix S = x^T x [K][K] (rcf_conv2d_wgrad_bf16 with dy = x).  W [N][K] is the fp32 master weight; every routine rounds it to
 * bf16 as the conv kernels do.
 *   forward   rcf_fold_fwd_f32:       P = W (S - A1 A1^T / rows_local) [N][K] + the N local means of z behind it (N K + N floats, kept
 *                                     for the backward pass; the Gram matrix is centred in fp64 BEFORE the fp32 contraction, so the
 *                                     variance w . P / n carries no mean^2 to cancel), sums = [sum z | sum z^2] (fp64 [2N]; SyncBN
 *                                     all-reduces them like any other statistics) and / or the finalized norm
 *             rcf_fold_finalize_f32:  mean, invstd, running statistics, num_batches_tracked as rcf_bn_finalize_f32, plus the
 *                                     folded constants scale = gamma invstd, shift = beta - mean scale (after the all-reduce)
 *             rcf_conv2d_fwd_affine_bf16:  y = [relu](conv(x, w) * scale[c] + shift[c] [+ residual]) (bf16; whole column tiles:
 *                                     Cout % 64 == 0 and Cout in {64, 128} or Cout % 256 == 0; stride 1)
 *   backward  g = dy under the ReLU mask + its column sums: rcf_relu_mask_colsum_bf16 (one pass, in place allowed), or
 *             rcf_conv2d_dgrad_masked_bf16 when a data gradient is the LAST writer of dy: dx = mask_src > 0 ? dgrad (+ dx) : 0
 *             with colsums [2 Cin] (sum | sum of squares) from its epilogue
 *             G = g^T x [N][K]: rcf_conv2d_wgrad_bf16(x, g)
 *             rcf_fold_bwd_sums_f32:     sums2 = [sum g | sum g zhat] (fp64 [2N]) from G, W, the column sums of g
 *             rcf_fold_wg_bf16:          wg_t = the data gradient's weight operand (bf16, rcf_conv_weight_bf16 layout, transpose = 1) of
 *                                     scale[c] W[c][k] -- forward constants only, so g Wg^T may run beside the G product
 *             rcf_fold_bwd_prepare_f32:  dW += ..., dgamma += sums2_local[N + c], dbeta += sums2_local[c] (sums2_local NULL = sums2),
 *                                     negT (bf16, transpose = 0 layout of the K -> K weight -T), c0 [K]
 *             dx = rcf_conv2d_dgrad_bf16(g, w_pairs_t = wg_t) then rcf_conv2d_fwd_bf16(x, negT, bias = c0, beta = 1)
 * N % 64 == 0 and K % 64 == 0. */
typedef struct rcf_fold_finalize {
    double count;                 /* rows the statistics cover */
    float eps, momentum;
    const float *gamma, *beta;    /* the norm's affine parameters [C] */
    float *mean, *invstd;         /* out [C] */
    float *scale, *shift;         /* out [C]: gamma invstd, beta - mean gamma invstd */
    float *running_mean, *running_var;   /* may be NULL */
    long long *num_batches_tracked;      /* may be NULL */
} rcf_fold_finalize;
size_t rcf_fold_fwd_scratch_bytes(int N, int K);
/* A1: fp64 [K] (the first half of rcf_bn_stats_mp's sums of x); rows_local: the rows S and A1 were summed over (this rank's).
 * P: N K + N floats.  fin != NULL: the statistics are local (fin->count == rows_local), the norm is finalized by the same launch
 * (sums may be NULL); fin == NULL: sums only.  N % 64 == 0, K % 64 == 0. */
int rcf_fold_fwd_f32(const float *S, const double *A1, const float *W, float *P, double *sums, const rcf_fold_finalize *fin,
                     void *scratch, size_t scratch_bytes, double rows_local, int N, int K, void *stream);
int rcf_fold_finalize_f32(const double *sums, int C, const rcf_fold_finalize *fin, void *stream);
/* relu_bits (optional, with relu): receives the sign bits of y, one 64-bit wavefront ballot per (tile, wave, 32 x 32 sub-tile,
 * accumulator register) -- rcf_conv_relu_bits_bytes(rows, Cout) bytes, 1/16 of y; only a launch over the SAME [rows][Cout]
 * tensor can read them back (rcf_conv2d_dgrad_masked_bf16's mask_bits). */
size_t rcf_conv_relu_bits_bytes(long rows, int C);
int rcf_conv2d_fwd_affine_bf16(const void *x, const void *w_bf16, const float *scale, const float *shift, const void *residual,
                               int res_pitch, int relu, void *y, void *relu_bits, const rcf_conv_shape *s, void *stream);
size_t rcf_relu_mask_colsum_bf16_workspace_bytes(long rows, int C);
int rcf_relu_mask_colsum_bf16(const void *dy, int dy_pitch, const void *y, int y_pitch, void *g, int g_pitch, long rows, int C,
                              double *colsums, void *workspace, size_t workspace_bytes, void *stream);
size_t rcf_conv2d_dgrad_masked_bf16_workspace_bytes(const rcf_conv_shape *s);
/* the mask: mask_src (the bf16 ReLU output itself, [rows][Cin] with mask_pitch) or mask_bits (its sign bits as
 * rcf_conv2d_fwd_affine_bf16 wrote them: 1/16 of the bytes; mask_src may then be NULL) */
int rcf_conv2d_dgrad_masked_bf16(const void *dy, const void *w_t_bf16, void *dx, const rcf_conv_shape *s, int beta,
                                 const void *mask_src, int mask_pitch, const void *mask_bits, double *colsums, void *workspace,
                                 size_t workspace_bytes, void *stream);
int rcf_fold_bwd_sums_f32(const float *G, const float *W, const double *colsums, const float *mean, const float *invstd,
                          double *sums2, int N, int K, void *stream);
size_t rcf_fold_bwd_scratch_bytes(int N, int K);
int rcf_fold_wg_bf16(const float *W, const float *scale, void *wg_t_bf16, int N, int K, void *stream);
/* P: rcf_fold_fwd_f32's (N K + N floats) */
int rcf_fold_bwd_prepare_f32(const float *G, const float *P, const double *A1, const float *W, const double *sums2,
                             const double *sums2_local, double count, const float *mean, const float *invstd, const float *gamma,
                             float *dW, float *dgamma, float *dbeta, void *negT_bf16, float *c0, void *scratch,
                             size_t scratch_bytes, int N, int K, void *stream);

/* dw (fp32, [Cout][R][S][Cin]) (+)= sum over pixels of dy (bf16) * x (bf16); deterministic split-K through the workspace */
size_t rcf_conv2d_wgrad_bf16_workspace_bytes(const rcf_conv_shape *s, const rcf_conv_region *region);
int rcf_conv2d_wgrad_bf16(const void *x, const void *dy, float *dw, const rcf_conv_shape *s, const rcf_conv_region *region,
                          int beta, void *workspace, size_t workspace_bytes, void *stream);
/* ---- batch norm (training), fused ReLU / residual add / Dropout2d channel scale -----------------
 * Replaces (Sync)BatchNorm in models/resnet.py:159-162 and mmcv ConvModule's norm, the ReLU and
 * `out += identity` of models/resnet.py:268-300, and nn.Dropout2d of models/decode_head.py:84-85.
 * x is [rows][C] with pixel pitch `pitch` (rows = N*H*W).
 * stats: per-channel sum and sum of squares accumulated in fp64 -> sums[2*C] (doubles, [sum | sumsq]).
 * A data-parallel run all-reduces `sums` (and the row count) between stats and finalize: that IS SyncBN. */
size_t rcf_bn_stats_workspace_bytes(long rows, int C);
int rcf_bn_stats_f32(const float *x, long rows, int C, int pitch, double *sums, void *workspace,
                     size_t workspace_bytes, void *stream);
/* mean/invstd (biased var, eps) from sums over `count` rows; running stats updated with `momentum`
 * and the unbiased variance; running_* may be NULL. */
int rcf_bn_finalize_f32(const double *sums, double count, int C, float eps, float momentum, float *mean,
                        float *invstd, float *running_mean, float *running_var, void *stream);
/* y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ) [* chan_scale[n][c]]
 * rows_per_image only matters with chan_scale (Dropout2d keep-mask / keep-prob, NULL = off). */
int rcf_bn_apply_f32(const float *x, int x_pitch, const float *residual, int r_pitch, float *y, int y_pitch,
                     long rows, int C, const float *mean, const float *invstd, const float *gamma,
                     const float *beta, int relu, const float *chan_scale, long rows_per_image,
                     unsigned char *relu_mask, unsigned *amax_out, void *stream);
/* amax_out (may be NULL; here and in rcf_bn_bwd_apply_f32): amax_out[0] = max(amax_out[0], bits(max |output|)), the
 * operand range of the convs that read the output next (rcf_conv_shape), without another pass over it.
 * relu_mask (may be NULL): with relu, one byte per 4 channels [rows][C/4], bit e = output channel 4j+e was positive
 * before the clamp.  The backward kernels take it in place of y (1/16 of the bytes of re-reading y). */
/* eval-mode BN: same kernel with mean=running_mean, invstd from running_var */
int rcf_bn_invstd_from_var_f32(const float *var, int C, float eps, float *invstd, void *stream);
/* backward.  g = dy [* chan_scale] [* (y>0)];  sums2 = [sum g | sum g*xhat] (fp64).
 * After a data-parallel all-reduce of sums2:  dx = gamma*invstd*(g - sum_g/count - xhat*sum_gx/count) with the
 * GLOBAL sums and count; dgamma += sum_gx, dbeta += sum_g from sums2_local (this rank's sums before the
 * all-reduce; NULL = sums2) because the gradient all-reduce adds the ranks later; and (if dres != NULL)
 * dres (+)= g  (res_beta 0/1). */
int rcf_bn_bwd_reduce_f32(const float *dy, int dy_pitch, const float *x, int x_pitch, const float *y,
                          int y_pitch, long rows, int C, const float *mean, const float *invstd, int relu,
                          const unsigned char *relu_mask, const float *chan_scale, long rows_per_image,
                          double *sums2, void *workspace, size_t workspace_bytes, void *stream);
int rcf_bn_bwd_apply_f32(const float *dy, int dy_pitch, const float *x, int x_pitch, const float *y, int y_pitch,
                         float *dx, int dx_pitch, float *dres, int dres_pitch, int res_beta, long rows, int C,
                         const float *mean, const float *invstd, const float *gamma, int relu,
                         const unsigned char *relu_mask, const float *chan_scale, long rows_per_image,
                         const double *sums2,
                         const double *sums2_local, double count, float *dgamma, float *dbeta,
                         unsigned *amax_out, void *stream);

/* Mixed-precision forms of the batch-norm family (the `_f32` entry points above are these with every code RCF_F32).
 * xdt: storage type of the conv-output side (x, dx); ydt: of the activation side (y, residual, dy, dres).
 * Supported (xdt, ydt): (F32, F32), (BF16, BF16), (F32, BF16) -- an fp32 conv output (the stem) normalised into bf16
 * activations.  The reference runs these layers under torch autocast (configs/rcf_stv2/rcf_stage1.yaml:57-60). */
int rcf_bn_stats_mp(const void *x, int xdt, long rows, int C, int pitch, double *sums, void *workspace,
                    size_t workspace_bytes, void *stream);
/* out (+)= relu_mask ? dy : 0 (bit e of byte [row][c/4] = channel 4 (c/4) + e was positive, as rcf_bn_apply_mp writes it): the
 * gradient a join relu(bn3(x) + identity) passes to its identity branch (models/resnet.py:296-300), for callers that did not
 * let rcf_bn_bwd_apply_mp write it (dres = NULL) and cannot take it as the addend of rcf_conv2d_dgrad_add_f32. */
int rcf_relu_mask_copy_mp(const void *dy, int dt, int dy_pitch, const unsigned char *relu_mask, void *out, int out_pitch,
                          long rows, int C, int beta, void *stream);
/* `flags` of the three streaming passes (per call; 0 = the library's own choices):
 *   RCF_BN_SWEEP_OFF / RCF_BN_SWEEP_ALWAYS   row order (csrc/bn.hip struct Sweep): by default tensors of 192 MB and more are walked
 *       in eight bands (the conv kernels' XCD bands), the forward apply and the backward reduction downwards, the backward apply
 *       upwards -- a kernel starts in what its producer touched last, which the 256 MB Infinity Cache still holds; OFF: front to
 *       back; ALWAYS: banded on every tensor of 8192 rows and more (tests).  Element-wise outputs are bit-identical.
 *   RCF_BN_Y_PLANES_ONLY (apply, with planes_out)   y itself is not written (its only consumers are convs); y may be NULL.
 *   RCF_BN_DX_PLANES (bwd_apply)   dx is written as fp16 pair planes instead of fp32 (dx_pitch == C).
 * Pair planes (RCF_CONV_X_PLANES / RCF_CONV_DY_PLANES above): fp32 tensors only, C % 8 == 0, no chan_scale.  planes_out (apply):
 * [rows][h: C fp16 | m: C fp16] of y * 2^k; the scale comes from an upper bound of |y| (resp. |dx|) that every workgroup derives
 * from the per-channel constants, amax_x = the range of x (the conv epilogue's by-product) and amax_res / amax_dy = the ranges
 * of the residual / of dy BEFORE reading an element; the bound's raw bits are left in *amax_out, which is what the consuming
 * convs take as amax_x / amax_dy. */
#define RCF_BN_SWEEP_OFF 0x1u
#define RCF_BN_SWEEP_ALWAYS 0x2u
#define RCF_BN_Y_PLANES_ONLY 0x4u
#define RCF_BN_DX_PLANES 0x8u
int rcf_bn_apply_mp(const void *x, int xdt, int x_pitch, const void *residual, int r_pitch, void *y, int ydt, int y_pitch,
                    long rows, int C, const float *mean, const float *invstd, const float *gamma, const float *beta,
                    int relu, const float *chan_scale, long rows_per_image, unsigned char *relu_mask,
                    unsigned *amax_out, void *planes_out, const unsigned *amax_x, const unsigned *amax_res, unsigned flags,
                    void *stream);
/* rcf_bn_apply_mp whose residual is itself a conv output in front of a finalized train-mode batch norm WITHOUT ReLU (a stage's
 * downsample branch: models/resnet.py:293-294 `identity = self.downsample(x)`, `out += identity`): res_norm's four [C] vectors
 * normalise the residual on the fly -- (r - mean) * invstd * gamma + beta, the apply pass's own operations and order, rounded to
 * the storage type as the stored tensor would have been -- so that norm's apply pass never runs.  amax_res is then the range of
 * the RAW residual.  res_norm == NULL: rcf_bn_apply_mp. */
typedef struct rcf_bn_res_norm {
    const float *mean, *invstd, *gamma, *beta;
} rcf_bn_res_norm;
int rcf_bn_apply_res_mp(const void *x, int xdt, int x_pitch, const void *residual, int r_pitch, const rcf_bn_res_norm *res_norm,
                        void *y, int ydt, int y_pitch, long rows, int C, const float *mean, const float *invstd,
                        const float *gamma, const float *beta, int relu, const float *chan_scale, long rows_per_image,
                        unsigned char *relu_mask, unsigned *amax_out, void *planes_out, const unsigned *amax_x,
                        const unsigned *amax_res, unsigned flags, void *stream);
/* The join of a stage's first block together with its downsample norm in the backward pass (models/resnet.py:293-296): both
 * norms' backward is taken of g = dy under the JOIN's sign bits, against conv3's output x and the downsample conv's output x2.
 * rcf_bn_bwd_reduce2_mp: ONE reduction -> sums4 [4C] = [sum g | sum g xhat | sum g | sum g xhat2] (each norm's [2C] pair is
 * contiguous: what rcf_bn_bwd_reduce_mp would give it, bit for bit); workspace: 2 x rcf_bn_stats_workspace_bytes(rows, C).
 * rcf_bn_bwd_apply2_mp: ONE apply pass -> dx and second->dx (both fp32 / bf16, or with RCF_BN_DX_PLANES both as fp16 pair
 * planes, each with its own bound in its own amax_out), the parameter gradients of both norms; dy and the sign bits are read
 * once.  Same element operations in the same order as two rcf_bn_bwd_apply_mp calls with relu = 1 and this mask. */
typedef struct rcf_bn_bwd_second {
    const void *x;                 /* the second norm's input [rows][C], storage type of x */
    int x_pitch;
    void *dx;                      /* its input gradient */
    int dx_pitch;
    const float *mean, *invstd, *gamma;
    const double *sums2;           /* [2C] (global under SyncBN) */
    const double *sums2_local;     /* [2C] this rank's (parameter gradients); NULL = sums2 */
    float *dgamma, *dbeta;         /* accumulated into; may be NULL */
    unsigned *amax_out;            /* range of dx (fp32 / bf16: may be NULL) or, with planes, the bound they are scaled by */
    const unsigned *amax_x;        /* planes: range of x */
} rcf_bn_bwd_second;
int rcf_bn_bwd_reduce2_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, const void *x2, int x2_pitch,
                          long rows, int C, const float *mean, const float *invstd, const float *mean2, const float *invstd2,
                          const unsigned char *relu_mask, double *sums4, void *workspace, size_t workspace_bytes, unsigned flags,
                          void *stream);
int rcf_bn_bwd_apply2_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, void *dx, int dx_pitch,
                         long rows, int C, const float *mean, const float *invstd, const float *gamma,
                         const unsigned char *relu_mask, const double *sums2, const double *sums2_local, double count,
                         float *dgamma, float *dbeta, unsigned *amax_out, const unsigned *amax_x, const unsigned *amax_dy,
                         const rcf_bn_bwd_second *second, unsigned flags, void *stream);
int rcf_bn_bwd_reduce_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, const void *y,
                         int y_pitch, long rows, int C, const float *mean, const float *invstd, int relu,
                         const unsigned char *relu_mask, const float *chan_scale, long rows_per_image, double *sums2,
                         void *workspace, size_t workspace_bytes, unsigned flags, void *stream);
int rcf_bn_bwd_apply_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, const void *y,
                        int y_pitch, void *dx, int dx_pitch, void *dres, int dres_pitch, int res_beta, long rows, int C,
                        const float *mean, const float *invstd, const float *gamma, int relu,
                        const unsigned char *relu_mask, const float *chan_scale, long rows_per_image,
                        const double *sums2, const double *sums2_local, double count, float *dgamma, float *dbeta,
                        unsigned *amax_out, const unsigned *amax_x, const unsigned *amax_dy, unsigned flags, void *stream);
int rcf_colsum_mp(const void *x, int xdt, long rows, int C, int pitch, float *out, int beta, void *workspace,
                  size_t workspace_bytes, void *stream);

/* ---- pooling / resize / layout ------------------------------------------------------------------
 * MaxPool2d(3,2,1): models/resnet.py:577.  argmax: uint8 window position (r*3+s), first max wins. */
int rcf_maxpool3x3s2_fwd_f32(const float *x, float *y, uint8_t *argmax, int N, int H, int W, int C, int Ho,
                             int Wo, void *stream);
int rcf_maxpool3x3s2_bwd_f32(const float *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C,
                             int Ho, int Wo, void *stream);
/* bilinear resize, NHWC, align_corners 0/1 (mmseg `resize`: models/decode_head.py:157-163,
 * models/rcf_model.py:213-220).  bwd is a gather over the contributing outputs (no atomics). */
int rcf_resize_bilinear_nhwc_fwd_f32(const float *x, int x_pitch, float *y, int y_pitch, int N, int Hi, int Wi,
                                     int Ho, int Wo, int C, int align_corners, void *stream);
int rcf_resize_bilinear_nhwc_bwd_f32(const float *dy, int dy_pitch, float *dx, int dx_pitch, int beta, int N,
                                     int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void *stream);
/* planar NCHW bilinear resize (ground-truth flows 480x854 -> mask size, models/rcf_model.py:438-442) */
/* the same restricted to the border frame of thickness `frame` (fine-resolution pixels): fwd writes only the frame's
 * output pixels, bwd takes dy as zero off the frame (and does not read it there).  See rcf_conv_region. */
int rcf_resize_bilinear_nhwc_fwd_frame_f32(const float *x, int x_pitch, float *y, int y_pitch, int N, int Hi, int Wi,
                                           int Ho, int Wo, int C, int align_corners, int frame, void *stream);
int rcf_resize_bilinear_nhwc_bwd_frame_f32(const float *dy, int dy_pitch, float *dx, int dx_pitch, int beta, int N,
                                           int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int frame,
                                           void *stream);
int rcf_resize_bilinear_nchw_f32(const float *x, float *y, int planes, int Hi, int Wi, int Ho, int Wo,
                                 int align_corners, void *stream);
/* NCHW [N,C,H,W] -> NHWC [N,H,W,Cpad] (channels >= C zero-filled) and back (drops the padding) */
int rcf_nchw_to_nhwc_f32(const float *x, float *y, int N, int C, int H, int W, int Cpad, void *stream);
int rcf_nhwc_to_nchw_f32(const float *x, int x_pitch, float *y, int N, int C, int H, int W, void *stream);
/* strided 2-D copy: dst[r*dpitch + c] (+)= src[r*spitch + c], c < C (concat / pair-concat / slices) */
int rcf_copy2d_f32(const float *src, long spitch, float *dst, long dpitch, long rows, int C, int beta,
                   void *stream);
/* n0 x n1 such copies in one launch: copy (i0, i1) reads src + i0*sb0 + i1*sb1 and writes dst + i0*db0 + i1*db1
 * (element offsets, multiples of 4, may be negative) -- the frames-of-a-pair channel concat of models/rcf_model.py:325 */
int rcf_copy2d_batched_f32(const float *src, long spitch, long sb0, long sb1, float *dst, long dpitch, long db0, long db1,
                           long rows, int C, int beta, int n0, int n1, void *stream);
/* dense NHWC [N,H,W,C]: inside = src on the rectangle (0 elsewhere), outside = src off the rectangle (0 on it);
 * either output may be NULL.  Splits a gradient into its interior / border-band parts (see rcf_conv_region). */
int rcf_split_rect_f32(const float *src, float *inside, float *outside, int N, int H, int W, int C, int y0, int x0,
                       int h, int w, void *stream);
/* Mixed-precision forms (dt = RCF_F32 / RCF_BF16 storage of every tensor argument; `frame` 0 = the whole tensor; -1 = the whole
 * tensor on the general kernels even where the exact-2x forms apply -- a thread there makes the 2 x 2 outputs of one source pixel
 * from 9 loads instead of 16, bit-identical -- for tests) */
int rcf_maxpool3x3s2_fwd_mp(const void *x, void *y, int dt, uint8_t *argmax, int N, int H, int W, int C, int Ho, int Wo,
                            void *stream);
int rcf_maxpool3x3s2_bwd_mp(const void *dy, const uint8_t *argmax, void *dx, int dt, int N, int H, int W, int C, int Ho,
                            int Wo, void *stream);
int rcf_resize_bilinear_nhwc_fwd_mp(const void *x, int x_pitch, void *y, int y_pitch, int dt, int N, int Hi, int Wi, int Ho,
                                    int Wo, int C, int align_corners, int frame, void *stream);
int rcf_resize_bilinear_nhwc_bwd_mp(const void *dy, int dy_pitch, void *dx, int dx_pitch, int dt, int beta, int N, int Hi,
                                    int Wi, int Ho, int Wo, int C, int align_corners, int frame, void *stream);
int rcf_copy2d_batched_mp(const void *src, long spitch, long sb0, long sb1, void *dst, long dpitch, long db0, long db1, int dt,
                          long rows, int C, int beta, int n0, int n1, void *stream);
/* source and destination in their own storage types: strided copy, or the fp32 <-> bf16 cast */
int rcf_copy2d_mp(const void *src, int sdt, long spitch, void *dst, int ddt, long dpitch, long rows, int C, int beta,
                  void *stream);
int rcf_split_rect_mp(const void *src, void *inside, void *outside, int dt, int N, int H, int W, int C, int y0, int x0, int h,
                      int w, void *stream);
/* column sums of [rows][C] (pitch) in fp64 -> out[C] (+)= (bias gradients of conv_seg) */
int rcf_colsum_f32(const float *x, long rows, int C, int pitch, float *out, int beta, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ---- backward warp / occlusion / photometric residual -------------------------------------------
 * utils/warp_utils.py:84-94 (flow_warp; pad 0 = 'border', 1 = 'zeros'), :107-113 + :27-81
 * (get_occu_mask_backward), :97-104 (get_occu_mask_bidirection), models/amd/flow_loss.py:15-29 with
 * models/amd/loss_blocks.py:46-65 (L1 + SSIM photometric loss).  Planar NCHW fp32 as the reference. */
/* pad_mode: 0 border, 1 zeros.  RGB / border-mode calls take the tile kernels (lane = x, dword taps); pad_mode | RCF_WARP_PER_PIXEL
 * keeps such a call on the per-pixel kernels every other call takes (bit-identical results; a per-call choice for tests) */
#define RCF_WARP_PER_PIXEL 0x100
int rcf_flow_warp_f32(const float *x, const float *flow, float *out, int B, int C, int H, int W, int pad_mode,
                      void *stream);
/* grads of flow_warp w.r.t. x (atomic scatter; dx must be zero-filled or hold a running sum) and flow */
int rcf_flow_warp_bwd_f32(const float *x, const float *flow, const float *dout, float *dx, float *dflow, int B,
                          int C, int H, int W, int pad_mode, void *stream);
/* occ[B,1,H,W] = (clamp(splat(flow21),0,1) < th); scratch: B*H*W floats */
int rcf_occu_mask_backward_f32(const float *flow21, float *occ, float th, float *scratch, int B, int H, int W,
                               void *stream);
int rcf_occu_mask_bidirection_f32(const float *flow12, const float *flow21, float *occ, float scale, float bias,
                                  int B, int H, int W, void *stream);
/* fused warp + occlusion-masked L1: out[0] = sum |im1 - warp(im2,flow)| * occ, out[1] = sum occ (fp64) */
int rcf_warp_l1_residual_f32(const float *im1, const float *im2, const float *flow, const float *occ, double *out,
                             int B, int C, int H, int W, int pad_mode, void *stream);
/* loss_photomatric: (w_l1*mean(|a-b|*occ) + w_ssim*mean(SSIM(b*occ,a*occ))) / mean(occ) -> out[0] (fp32).
 * scratch: 4 doubles. */
int rcf_photometric_loss_f32(const float *im, const float *recon, const float *occ, float w_l1, float w_ssim,
                             float *out, double *scratch, int B, int C, int H, int W, void *stream);

/* ---- dense CRF (permutohedral mean-field) -------------------------------------------------------
 * Replaces torchcrf_cpp.crf_soft / crf_hard (tools/torchCRF/src/torchcrf.cu:106-149), batched over
 * frames, persistent caller-owned workspace instead of 12 cudaMalloc/cudaFree per frame.
 *   rgb   : u8 [batch,H,W,3]         unary : f32 [batch,H*W,2] energies (label minor)
 *   out   : i16 [batch,H,W] MAP      q_out : optional f32 [batch,H*W,2] final marginals (NULL = skip)
 *   nvert : optional int32 [batch,2] lattice vertex counts (smoothness, appearance)
 * A potential is active when its weight and sigma are > 0 (torchcrf.cu:28,41). */
size_t rcf_crf_workspace_bytes(int W, int H, int batch);
int rcf_crf_soft(const uint8_t *rgb, const float *unary, int W, int H, int batch, float scomp_smooth,
                 float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, int iters, int16_t *out_map,
                 float *q_out, int32_t *nvert, void *workspace, size_t workspace_bytes, void *stream);
/* normalization 0 = rcf_crf_soft; 1 = the symmetric kernel normalisation N^1/2 K N^1/2, N = diag(1 / (K 1 + 1e-20)), of
 * pydensecrf's DenseCRF2D (NORMALIZE_SYMMETRIC, the default of addPairwiseBilateral): the CPU post-processor
 * tools/pydenseCRF/crf.py:58-89 and models/crf_head.py:62-91.  Exactly one potential may be active with 1. */
int rcf_crf_soft_ex(const uint8_t *rgb, const float *unary, int W, int H, int batch, float scomp_smooth, float sxy_smooth,
                    float scomp_app, float sxy_app, float srgb_app, int iters, int normalization, int16_t *out_map,
                    float *q_out, int32_t *nvert, void *workspace, size_t workspace_bytes, void *stream);
/* rcf_crf_soft_ex on FLOAT colour features f32 [batch,H,W,3]: torchcrf_cpp.crf_soft accepts rgbFeat of any dtype and converts
 * it to float unrounded (tools/torchCRF/src/torchcrf.cu:84-85); RCF itself passes uint8 (models/crf_head.py:43-55).  Always
 * the array-of-keys lattice build (the RCF_CRF_BUILD_* bits are ignored). */
int rcf_crf_soft_f32(const float *rgbf, const float *unary, int W, int H, int batch, float scomp_smooth, float sxy_smooth,
                     float scomp_app, float sxy_app, float srgb_app, int iters, int normalization, int16_t *out_map,
                     float *q_out, int32_t *nvert, void *workspace, size_t workspace_bytes, void *stream);
int rcf_crf_hard(const uint8_t *rgb, const int16_t *label, int W, int H, int batch, float scomp_smooth,
                 float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, float confidence, int iters,
                 int16_t *out_map, float *q_out, int32_t *nvert, void *workspace, size_t workspace_bytes,
                 void *stream);
/* lattice build, OR-ed into rcf_crf_soft_ex's `normalization` (a per-call choice; results are identical): default = packed
 * 64-bit keys + block-local de-duplication when the key coordinates fit 12 bits; ARRAY = always the array-of-keys build;
 * SMALL_TABLE = packed build whose first-attempt table is tiny, so every frame takes the overflow path (tests) */
#define RCF_CRF_BUILD_ARRAY 0x100
#define RCF_CRF_BUILD_SMALL_TABLE 0x200
/* SORT = the appearance lattice by ONE device-wide radix sort of all frames' (packed key, entry) pairs + run heads + scan
 * instead of the hash table (the sort / unique / scan form of SURVEY section 7 step 6): vertices numbered in key order, neighbour
 * search as a merge.  Slower than the packed build on natural frames (a few 10^4 vertices per frame), faster on noise-like
 * ones (10^6 vertices): what the caller picks when the previous call's vertex counts were high (rcf_amd.crf.CRFHead).
 * Needs keys that fit 12 bits per coordinate and batch <= 16; otherwise the default build runs. */
#define RCF_CRF_BUILD_SORT 0x300
/* the six blur passes of a filter as six launches (one per lattice axis) instead of three (two axes per launch, the first
 * axis's values recomputed on the fly for the three vertices the second reads: same operations, same bits) -- tests and A/B */
#define RCF_CRF_BLUR_SEQUENTIAL 0x400
/* The splat of a filter pass.  Default: frames built by the packed build whose 16 x 16 pixel tiles share most of their vertices
 * (natural images) take the TILE splat -- a scatter into the tile's short vertex list in LDS, one 64-bit atomic per distinct
 * vertex of the tile and channel, no CSR list built -- the others the gather over the CSR lists.  Sums are in 2^-40 fixed point
 * either way: identical bits.  SPLAT_GATHER: always the gather (A/B, tests); SPLAT_TILES: the tile splat for every frame the
 * packed build made, whatever its content (tests). */
#define RCF_CRF_SPLAT_GATHER 0x4000
#define RCF_CRF_SPLAT_TILES 0x8000
/* With ONE potential active, the slice of a filter pass hands a tile-mode frame's new marginals straight to the next pass's
 * per-tile sums (one kernel reads the entries' weights and list positions once; the intermediate marginals are never written).
 * SLICE_SPLAT_SEPARATE: slice and per-tile sums as two kernels, as with two potentials (tests, A/B; identical bits). */
#define RCF_CRF_SLICE_SPLAT_SEPARATE 0x10000
/* CRFHead pre-processing (models/crf_head.py:33-37,43-55,95-98): normalised NCHW image -> u8 HWC;
 * soft mask -> u8 quantisation -> unary energies.  scratch: batch uint32 (per-frame max). */
int rcf_crf_prepare(const float *img_nchw, const float *mask, const float *mean3, const float *std3,
                    int unstandardize, float crf_scale, uint8_t *rgb_out, float *unary_out, uint32_t *scratch,
                    int batch, int H, int W, void *stream);

/* ---- flow-aggregation head (relaxed common fate) + loss tail ------------------------------------
 * models/flow_aggregation_head_with_residual.py:235-310 (aggregate), :164-233 (per-segment affine
 * least squares), :312-399 (forward, L1 / robust loss); softmax + double-softmax entropy of
 * models/rcf_model.py:433-434,376-378; asymmetric clamped MSE against pl / crf targets :380-408.
 * "Direction image" n = 2*b + d: d = 0 forward (frame-0 masks, fw flow, residual channels [0,2C)),
 * d = 1 backward (frame-1 masks, bw flow, residual channels [2C,4C)); P = h*w.
 *   logits   NHWC [2B][P][logits_pitch]         (conv_seg output of decode_head2)
 *   feat     NHWC [2B][P][64]                   (flow_feat_before_agg output, LeakyReLU applied)
 *   residual NHWC [B][P][4C]                    (decode_head3 output resized to the mask size)
 *   W1 [64][64], b1 [64], W2 [2][64], b2 [2]    (flow_feat_after_agg Conv1d weights)
 *   targets  [2B][P] planar or NULL             (pl / crf masks at mask size)
 * The same workspace must be passed to prepare -> fwd -> bwd of one step. */
typedef struct {
    int B, C, h, w;          /* pairs, segments (mask_layer), mask size */
    int logits_pitch;
    int nf;                  /* must be 64 */
    int D;                   /* 0 free_residual, 2 free_residual_with_affine, 5 + quadratic */
    int robust;              /* outlier_robust_loss */
    int tanh_residual;       /* 1: scale*tanh(r/div) ; 0: plain r (residual_adjustment_scale == -1) */
    float eps, q;
    float clamp_t;           /* < 0: no clamp */
    float res_scale, div_coeff;
    float w_seg, w_entropy;
    int n_targets, target_channel;
    float t_wpos[2], t_wneg[2], t_weight[2], t_thresh[2];   /* t_thresh -1: soft target */
    /* models/compactness_head.py:14-57 (GWM compactness of one channel); weight 0 = off */
    float w_compact;
    int compact_channel;
    /* models/rcf_model.py:350-374 sharpen loss: mode 0 off, 1 KL to the sharpened (p^(1/T)) masks,
     * 2 object-aware hinge on |p_obj - max other| (object channel = target_channel) */
    float w_sharpen, t_sharpen;
    int sharpen_mode;
} rcf_flowhead_cfg;
size_t rcf_flowhead_workspace_bytes(const rcf_flowhead_cfg *c);
/* gt_fw / gt_bw [B][2][P] -> clamped flows (kept in the workspace) and flow4 NHWC [2B][P][4], the
 * zero-padded input of flow_feat_before_agg.0 */
int rcf_flowhead_prepare_f32(const rcf_flowhead_cfg *c, const float *gt_fw, const float *gt_bw, float *flow4,
                             void *workspace, size_t workspace_bytes, void *stream);
/* losses_out[8] = seg_fw, seg_bw, entropy, target0, target1, compactness, sharpen, 0 (unweighted means).  masks_out [2B][C][P] and
 * the four flow planes [2B][2][P] (overall, aggregated, residual adjustment, affine) are optional. */
int rcf_flowhead_fwd_f32(const rcf_flowhead_cfg *c, const float *logits, const float *feat, const float *residual,
                         const float *W1, const float *b1, const float *W2, const float *b2, const float *target0,
                         const float *target1, float *losses_out, float *masks_out, float *flow_pred,
                         float *flow_agg, float *flow_adj, float *flow_aff, void *workspace, size_t workspace_bytes,
                         void *stream);
/* gradients of grad_scale * (w_seg*seg + w_entropy*entropy + sum t_weight*target + w_compact*compactness +
 * w_sharpen*sharpen): dlogits (same layout as
 * logits, written), dresidual [B][P][4C] (written), dfeat [2B][P][64] = gradient w.r.t. the PRE-activation of
 * the second flow conv (written); dW1/db1/dW2/db2 accumulate. */
int rcf_flowhead_bwd_f32(const rcf_flowhead_cfg *c, const float *feat, const float *residual, const float *W1,
                         const float *W2, const float *target0, const float *target1, float grad_scale,
                         float *dlogits, float *dresidual, float *dfeat, float *dW1, float *db1, float *dW2,
                         float *db2, void *workspace, size_t workspace_bytes, void *stream);
/* dx = dy * (y > 0 ? 1 : slope), y = LeakyReLU output (n % 4 == 0) */
int rcf_lrelu_bwd_f32(const float *dy, const float *y, float *dx, long n, float slope, void *stream);

/* ---- optimiser / EMA ----------------------------------------------------------------------------
 * torch.optim.Adam with coupled weight decay (main.py:299-307) over one flat fp32 buffer;
 * EMA lerp of utils/model_utils.py:33-38 over flat buffers. */
int rcf_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                      void *stream);
int rcf_ema_update_f32(float *dest, const float *src, long n, float m, void *stream);
/* ... over ALL entries of a state dict in one launch (utils/model_utils.py:33-38 loops over them: ~500 launches of a few
 * microseconds per step for the EMA teacher, which the host cannot enqueue as fast as the GPU runs them -- 5.7 ms of idle GPU per
 * bf16 stage-2.1 step).  `chunks_dev`: device array of `count` chunks, one workgroup each; kind 0: fp32 dest = dest m + src (1 - m),
 * kind 1: int64 counters (num_batches_tracked), dest = trunc(float(dest) m + float(src) one_minus_m_counter) -- what torch's
 * int64 * python-float arithmetic gives the reference. */
typedef struct rcf_ema_chunk {
    void *dst;
    const void *src;
    long n;
    long kind;
} rcf_ema_chunk;
int rcf_ema_update_multi(const rcf_ema_chunk *chunks_dev, int count, float m, float one_minus_m_counter, void *stream);
int rcf_fill_f32(float *p, long n, float v, void *stream);
/* nn.Dropout2d's draw as a per-(sample, channel) scale (models/decode_head.py:84-87, models/fcn_head.py:142-147): out[i] = 0 with
 * probability p, 1 / (1 - p) otherwise, i < n = samples * channels; Philox4x32-10 keyed by `seed`, counter = i (reproducible
 * whatever the launch geometry).  The batch-norm apply pass of the head's last conv module multiplies it in (chan_scale). */
int rcf_dropout2d_scale_f32(float *out, long n, float p, unsigned long long seed, void *stream);

/* ---- evaluation metric ---------------------------------------------------------------
 * main.py:200-235 + utils/eval_utils.py:5-52,120-123: masks [B,C,h,w] (fp32 softmax) are resized to the annotation size
 * [H,W] (bilinear, align_corners=True), binarised (> pos_th; pos_th == -1 exactly: one-hot of the channel argmax) and compared with
 * the u8 annotation (255 foreground, 128 ignored, else background).  counts [B][C][3] uint64, zero-filled by the caller:
 * intersection, prediction area, label area of the foreground over the valid pixels; IoU = I / (P + L - I). */
int rcf_eval_iou_counts_f32(const float *masks, const uint8_t *ann, int B, int C, int h, int w, int H, int W, float pos_th,
                            unsigned long long *counts, void *stream);

/* ---- data transform on the device ------------------------------------------------------
 * dataset/transforms.py:884-924 `Transform` (Resize :170-237 -> RandomCrop :442-508 -> RandomFlip :249-306 ->
 * PhotoMetricDistortion :557-687 -> FlowTransform :825-848 / PLTransform :865-876 -> NumpyToTensor :793-808 ->
 * TorchNormalize :850-863) as one gather per output tensor.  The random decisions are drawn by the host in the
 * reference's order and passed per sample: */
typedef struct rcf_aug_params {
    int32_t rw, rh;            /* size of the resized frame (mmcv.rescale_size of the drawn scale) */
    int32_t crop_x, crop_y;    /* RandomCrop offset inside the resized frame (0,0 and out = resized size: no crop) */
    int32_t flip;              /* horizontal flip of the crop */
    int32_t ops;               /* bit 0 brightness, 1 contrast, 2 saturation, 3 hue, 4 contrast applied last (mode 0) */
    float beta;                /* brightness offset */
    float alpha_c, alpha_s;    /* contrast / saturation gains */
    float flow_sx, flow_sy;    /* FlowTransform(scale_flow): factors on (u, v); 1 when off */
    double hue_delta;          /* python float in the reference: the hue arithmetic is fp64 */
} rcf_aug_params;
/* frames u8 [B][I][H][W][3] (decoded RGB) -> out fp32 [I][B][3][oh][ow]: (u8/255 - mean)/std of the bilinearly resized
 * (cv2 8-bit fixed point), cropped, flipped, distorted frame.  params: device array [B]; mean3/std3: host arrays. */
int rcf_aug_frames_u8(const uint8_t *frames, int B, int I, int H, int W, const rcf_aug_params *params, float *out, int oh,
                      int ow, const float *mean3, const float *std3, void *stream);
/* seg fields (nearest resize, crop, flip -- the reference does not negate u under a flip): flows fp32 [B][K][H][W][2] ->
 * [K][B][2][oh][ow];  pseudo-label masks u8 [B][K][H][W] -> fp32 [K][B][oh][ow] = u8/255 */
int rcf_aug_flows_f32(const float *flows, int B, int K, int H, int W, const rcf_aug_params *params, float *out, int oh,
                      int ow, void *stream);
int rcf_aug_masks_u8(const uint8_t *masks, int B, int K, int H, int W, const rcf_aug_params *params, float *out, int oh,
                     int ow, void *stream);

/* ---- DINO ViT forward + soft NCut -----------------------------------------
 * models/dino_vit.py:110-167,176-276 (nn.Linear / attention products, LayerNorm eps 1e-6, softmax, GELU) and
 * tools/SemanticConstraintsAndMAA/semantic_constraints.py:21-75 (soft NCut value, Adam refinement of the mask). */
/* C[M][N] (pitch ldc) (+)= A[M][K] (pitch lda) . B[N][K]^T (pitch ldb) + bias[N]; act 0 none / 1 LeakyReLU / 2 GELU(erf).
 * Runs on the implicit-GEMM conv kernel (fp32-level error).  K, lda, ldb, ldc multiples of 4.
 * amax_a / amax_b (both or neither; may be NULL): operand ranges -> fp16 pairs instead of bf16 triples (rcf_conv_shape);
 * b_pairs (may be NULL): B already split by rcf_conv_weight_pairs_f32(B, N, K, 1, 1, amax_b, ...), needs ldb == K;
 * amax_out (may be NULL): amax_out[0] = max(amax_out[0], bits(max |C written|)) -- the range of whatever reads C next. */
int rcf_gemm_nt_f32(const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc, int M, int N,
                    int K, int act, float slope, int beta, const unsigned *amax_a, const unsigned *amax_b,
                    const void *b_pairs, unsigned *amax_out, void *stream);
/* batch0 x batch1 independent products in one launch (attention: images x heads); strides in elements */
int rcf_gemm_nt_batched_f32(const float *A, int lda, long a_s0, long a_s1, const float *B, int ldb, long b_s0, long b_s1,
                            float *C, int ldc, long c_s0, long c_s1, int batch0, int batch1, int M, int N, int K, int act,
                            float slope, int beta, void *stream);
/* fused multi-head attention forward (scores stay on chip): out[b*T+t][h*64+d] = softmax_j(scale q_t.k_j) v_j[d];
 * qkv [B*T][3*nh*64] (q | k | v) as produced by the fused qkv linear (models/dino_vit.py:122-133).  head_dim = 64. */
int rcf_attention_fwd_f32(const float *qkv, int ld_qkv, float *out, int ld_out, int B, int T, int nh, int head_dim,
                          float scale, const unsigned *amax_qkv, unsigned *amax_out, void *stream);
/* amax_qkv: max |qkv| (rcf_absmax_f32, or the amax_out of the qkv GEMM) -> fp16-pair arithmetic (q, k, v scaled by one
 * power of two, probabilities by 2^14, 3 partial products); NULL -> bf16 triples (6).  amax_out (may be NULL): the
 * range of `out`. */
int rcf_layernorm_f32(const float *x, int x_pitch, float *y, int y_pitch, long rows, int C, const float *gamma,
                      const float *beta, float eps, unsigned *amax_out, void *stream);
/* in place: row <- softmax(scale * row[0:n]); columns [n, pitch) are zeroed */
int rcf_softmax_rows_f32(float *s, long pitch, long rows, int n, float scale, void *stream);
/* dst[c][r] = src[r][c]; dst columns [rows, dpitch) zero-filled */
int rcf_transpose2d_f32(const float *src, long spitch, float *dst, long dpitch, int rows, int cols, void *stream);
int rcf_l2_normalize_rows_f32(const float *x, long x_pitch, float *y, long y_pitch, long rows, int C, void *stream);
/* gram[i][j] <- gram[i][j] > tau ? 1 : eps */
int rcf_affinity_threshold_f32(float *gram, long pitch, int n, float tau, float eps, void *stream);
int rcf_ncut_value_grad_f32(const float *affinity, long pitch, int n, const float *x, double *u, double *rowsum,
                            int compute_rowsum, float *grad, float *value_out, void *stream);
int rcf_clamp01_f32(float *x, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RCF_HIP_H */
