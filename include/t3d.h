/* t3d.h -- C ABI of the MI355X-native 3D-box keypoint-regression hot path.
 *
 * The reference (sovrasov/3d-object-detection.pytorch) is 100 % Python and has no
 * FFI of its own: every device op on its hot path is an ATen call issued from
 *   torchdet3d/models/mobilenetv3.py:110-166   (conv / BN / activation / SE)
 *   torchdet3d/builders/model_builder.py:96-146 (pool, per-class heads, cls head)
 *   torchdet3d/losses/regression_losses.py, builders/loss_builder.py (losses)
 *   torchdet3d/evaluation/metrics.py:10-37     (ADD / SADD / accuracy)
 * Each entry point below names the reference call site it replaces.  The
 * library (libt3d_hip.so, hand-written HIP for gfx950) is what the reference's
 * Python would bind with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes; all buffers are caller-allocated DEVICE memory,
 *     no ownership transfer, no hidden allocation or synchronisation;
 *   - every call only ENQUEUES on `stream` (a hipStream_t passed as void*);
 *   - activations are NHWC ("channels last"), storage dtype T3D_F32 or T3D_BF16,
 *     arithmetic always accumulates in fp32 (statistics in fp64);
 *   - channel counts are multiples of 8;
 *   - return 0 on success, a negative T3D_ERR_* otherwise.
 */
#ifndef T3D_H_
#define T3D_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { T3D_F32 = 0, T3D_BF16 = 1, T3D_F16 = 2 };    /* T3D_F16: INFERENCE forward only (round 4): t3d_stem_im2col[_u8], t3d_pwconv_fwd,
                                                         t3d_dwconv_fwd (k = 3), t3d_bn_apply, t3d_pool_fwd, t3d_pack_weight[s_batched] */
/* Weight layout flag of the 16-bit pointwise convolutions (round 4): `T3D_BF16 | T3D_W_FRAG` (or `T3D_F16 | ...`) as the dtype
 * argument of t3d_pwconv_fwd / t3d_pwconv_fwd_mat / t3d_pwconv_dgrad says that `w` / `wt` is the FRAGMENT-ORDER copy made by
 * t3d_pwconv_pack_frag (or by t3d_pack_weights_batched's `frag` / `frag_t` outputs) of the matrix the plain call takes.  The
 * deep-contraction kernel streams the fragments from L2 (csrc/pwconv_deep.hip; t3d_pwconv_wants_frag names its shapes), the
 * streaming kernel stages its weight chunk with one linear copy (csrc/pwconv_stream.hip); shapes neither of them takes return
 * T3D_ERR_UNSUPPORTED / T3D_ERR_ARG (the LDS-tiled fallback kernel reads the row-major matrix only). */
enum { T3D_W_FRAG = 0x100 };
enum { T3D_ACT_NONE = 0, T3D_ACT_RELU = 1, T3D_ACT_RELU6 = 2, T3D_ACT_HSWISH = 3 };
enum { T3D_OK = 0, T3D_ERR_ARG = -1, T3D_ERR_LAUNCH = -2, T3D_ERR_UNSUPPORTED = -3 };

/* How a consumer turns a stored RAW (pre-BatchNorm) tensor into its activated
 * input on load, so BN/activation/SE never cost a pass over HBM:
 *   u = scale[c]*x + shift[c]            (scale == NULL: u = x)
 *   se_after_act == 0:  a = act(u * se[b][c])      (expand layout, mobilenetv3.py:153-156)
 *   se_after_act == 1:  a = act(u) * se[b][c]      (no-expand layout, mobilenetv3.py:137-140)
 * se == NULL: no squeeze-excite factor. */
typedef struct {
  const float* scale;
  const float* shift;
  const float* se;
  int act;
  int se_after_act;
} t3d_prologue;

int t3d_version(void);

/* Depthwise k x k conv forward, k in {3,5}, stride in {1,2}, pad (k-1)/2.
 * Replaces nn.Conv2d(C,C,k,s,(k-1)//2,groups=C) + the following BatchNorm2d's
 * statistics pass (mobilenetv3.py:136-137,152-153).
 *   x [B,H,W,C] (dtype), w [C,k*k] fp32 (= the reference's [C,1,k,k]),
 *   y [B,Ho,Wo,C] raw output (dtype),
 *   stats  [2*C] fp64 or NULL: += sum(y), sum(y^2) per channel (must be zeroed by caller),
 *   gap_sum [B*C] fp32 or NULL: += sum over (Ho,Wo) of y per sample/channel (for SE). */
int t3d_dwconv_fwd(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y,
                   double* stats, float* gap_sum, int B, int H, int W, int C, int k, int stride,
                   void* stream);

/* BatchNorm training-mode statistics -> per-channel affine (mobilenetv3.py:113 etc.;
 * PyTorch defaults eps=1e-5, momentum=0.1, biased var to normalise, unbiased running var).
 *   stats [2*C] fp64 sums over `count` elements; scale = gamma*invstd, shift = beta - mean*scale.
 *   running_mean/var, num_batches_tracked may be NULL (not updated). */
int t3d_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                    void* stream);

/* Eval-mode BatchNorm folded to the same per-channel affine from the running estimates. */
int t3d_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream);

/* ... for every BatchNorm of a model in ONE launch (an inference forward otherwise issues ~50 of them): desc = n rows
 * of int64 {gamma, beta, running_mean, running_var, scale, shift, C}, a DEVICE array. */
int t3d_bn_eval_affine_batched(const long long* desc, int n, float eps, void* stream);

/* BatchNorm backward as a per-channel affine of two tensors, applied by the consumer on load:
 *   dy = alpha*dz + beta*y + gamma      (dz: gradient at the BN output, y: raw BN input)
 * alpha/gamma are [C], or [B*C] when per_sample != 0 (a squeeze-excite gate sits between). */
typedef struct {
  const float* alpha;
  const float* beta;
  const float* gamma;
  int per_sample;
} t3d_bnbwd;

/* Pointwise (1x1) conv forward as an MFMA GEMM.  Replaces nn.Conv2d(K,N,1) / nn.Linear(K,N)
 * + the statistics pass of the BatchNorm that follows (mobilenetv3.py:120-121,142-143,148-149,
 * 158-159,192-193).
 *   x [M,K] rows = NHWC pixels (M = B*H*W, HW = H*W for per-sample SE indexing),
 *   w [N,K] in the storage dtype, bias [N] fp32 or NULL, y [M,N] raw output,
 *   stats [2*N] fp64 or NULL: += sum(y), sum(y^2) per output channel (caller zeroes). */
int t3d_pwconv_fwd(int dtype, const void* x, const t3d_prologue* pro, const void* w, const float* bias,
                   void* y, double* stats, int M, int HW, int K, int N, void* stream);

/* Fragment-order weights (no reference counterpart: a layout of the weights nn.Conv2d holds, mobilenetv3.py:142-159).
 * `w` [rows, cols] in a 16-bit storage type, row-major -> out[((T*KS + ks)*64 + lg*16 + lc)*8 + j] = w[row(T, lc)][32 ks + 8 lg + j],
 * row(T, lc) = 32 (T >> 1) + 8 (lc >> 2) + 4 (T & 1) + (lc & 3), T < 2 ceil(rows / 32), KS = ceil(cols / 32), zero past rows / cols:
 * the A operand of v_mfma_f32_16x16x32_{bf16,f16} for 16 output channels x 32 contraction elements is 1 KB contiguous (a wave
 * loads it with one fully coalesced instruction instead of 64 separate 16-byte requests), and the row permutation inside a
 * pair of tiles leaves every lane with 8 consecutive output channels of a 32-channel block.  t3d_pwconv_frag_bytes: size of
 * `out`.  t3d_pwconv_wants_frag(K, N): 1 where the layout pays -- the deep-contraction kernel's shapes (K >= 512 and an output
 * wider than the streaming kernel's LDS chunk) -- for a contraction over K into N output channels (forward: (K, N) of the
 * layer with `w`; data gradient: (N, K) with the copy of `wt`).  The flag itself is accepted for every shape the streaming
 * kernel takes as well (K, N multiples of 8, K <= 1920). */
int t3d_pwconv_frag_bytes(int rows, int cols);
int t3d_pwconv_wants_frag(int K, int N);
int t3d_pwconv_pack_frag(const void* w, void* out, int rows, int cols, void* stream);

/* Pointwise conv data gradient (autograd of the conv above + the surrounding BN/activation).
 *   dz [M,N], y [M,N]: gradient at / raw input of the BatchNorm after the conv, bb its backward affine;
 *   wt [K,N]: TRANSPOSED weight in the storage dtype;
 *   x_raw [M,K] + pro_in: raw tensor the forward conv read and how it was activated; if given the
 *     result is multiplied by act'(.) and sum(dx), sum(dx*x_raw) are accumulated into
 *     stats [2*K] fp64 (per channel) or ps_stats [B*K*2] fp32 (per sample, SE case);
 *   residual [M,K] or NULL is added (skip connection); dx [M,K]. */
int t3d_pwconv_dgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* wt,
                     const void* x_raw, const t3d_prologue* pro_in, const void* residual, void* dx,
                     double* stats, float* ps_stats, int M, int HW, int K, int N, void* stream);

/* Sum of reduction replicas for several tensors in one launch (the depthwise weight gradients of every layer finished
 * since the last call): desc = n rows of int64 {src, dst, count}, src [nrep][count] fp32 -> dst [count] fp32 (overwritten). */
int t3d_sum_replicas_batched(const long long* desc, int n, int nrep, void* stream);

/* Weight packing: fp32 master weight [rows,cols] -> storage dtype, optionally transposed to
 * [cols,rows] (the dgrad GEMM reads the transposed copy). */
int t3d_pack_weight(int dtype, const float* w, void* out, int rows, int cols, int transpose, void* stream);

/* Pointwise conv weight gradient: dw[N,K] (fp32, the reference's [N,K,1,1]) += dy^T * a, with
 * dy = bb(dz, y) and a = pro(x) recomputed on load; the caller zeroes dw once per step. */
int t3d_pwconv_wgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* x,
                     const t3d_prologue* pro, float* dw, int M, int HW, int K, int N, void* stream);

/* "y-free" backward of a pointwise conv whose input x [M,K] is a finished (materialised) bf16 tensor -- the expand
 * layer nn.Conv2d(K, N, 1) + nn.BatchNorm2d(N) of an inverted-residual block (models/mobilenetv3.py:148-149).
 * Because y = x W^T, the BatchNorm-backward affine dy = alpha*dz + beta*y + gamma never needs the wide tensor y:
 *   dx = [dz | x] Wcat^T + c,           Wcat = [alpha.W | W^T diag(beta) W] (bf16 [K][rup32(N)+rup32(K)]), c = gamma^T W
 *   dW += alpha.(dz^T x) + beta.(W (x^T x)) + gamma (1^T x)
 * Same sums as t3d_pwconv_dgrad / t3d_pwconv_wgrad, reassociated; bf16 storage only, per-channel bb only.
 *   _prep: wt [K,N] bf16 (the TRANSPOSED packed weights, as the data gradient reads them) -> wcat, cvec [K] fp32;
 *   _dgrad_yfree: x_raw / pro_in / residual / stats exactly as in t3d_pwconv_dgrad (x itself is the finished tensor the
 *     conv read; x_raw the raw tensor of its producer, for that producer's BatchNorm-backward sums); dx [M,K] bf16;
 *   _wgrad_yfree: needs t3d_set_workspace (returns T3D_ERR_UNSUPPORTED without it); dw [N,K] fp32 is accumulated. */
int t3d_pwconv_yfree_prep(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, int K, int N, void* stream);
int t3d_pwconv_dgrad_yfree(const void* dz, const void* x, const void* wcat, const float* cvec, const void* x_raw,
                           const t3d_prologue* pro_in, const void* residual, void* dx, double* stats, int M, int HW,
                           int K, int N, void* stream);
int t3d_pwconv_wgrad_yfree(const void* dz, const void* x, const t3d_bnbwd* bb, const void* w, float* dw, int M, int HW,
                           int K, int N, void* stream);

/* The same y-free backward with ONE pass over the wide gradient tensor (round 4): the pair above reads dz [M,N] twice, at the
 * same time from two streams; t3d_pwconv_bwd_yfree forms dx AND the partial tiles of [dz | x | 1]^T x from the same staged rows.
 *   _prep2: as _prep, plus wd -- the weights in the staged rows' column order, [16*ceil(K/16)][64*ceil((N+K+8)/64)] bf16,
 *     cleared once by the caller (only the non-zero entries are written each step);
 *   _bwd_yfree_scratch: RETURNS the bytes of `scratch` a launch of this shape needs (a query, not a status code; 0: shape not
 *     supported -- K <= 32, 64 < N + K + 8 <= 256);
 *   _bwd_yfree (the caller's main stream): dx [M,K] bf16 = [dz | x | 1] wd^T (+ residual), stats += sum(dx), sum(dx*x_raw)
 *     (x_raw / pro_in as in t3d_pwconv_dgrad; only a linear producer -- no activation -- is supported), partial tiles -> scratch;
 *   _wgrad_yfree_finish (any stream ordered behind it): dw [N,K] += the combined weight gradient, from `scratch`.
 * Replaces the autograd of nn.Conv2d(K, N, 1) + nn.BatchNorm2d(N) (models/mobilenetv3.py:148-149), as the pair does. */
int t3d_pwconv_yfree_prep2(const void* wt, const t3d_bnbwd* bb, void* wcat, float* cvec, void* wd, int K, int N, void* stream);
/* _bwd_yfree_w (round 5): t3d_pwconv_bwd_yfree with the weight rows `wd` built in the launch's own prologue from wt [K,N] (the
 * transposed bf16 weights) and the BatchNorm-backward coefficients `bb` (or the pending finalize request for them): the same
 * numbers bit for bit, without the t3d_pwconv_yfree_prep2 launch in front of it on the critical stream. */
int t3d_pwconv_bwd_yfree_w(const void* dz, const void* x, const void* wt, const t3d_bnbwd* bb, const void* x_raw,
                           const t3d_prologue* pro_in, const void* residual, void* dx, double* stats, void* scratch,
                           long long scratch_bytes, int M, int HW, int K, int N, void* stream);
int t3d_pwconv_bwd_yfree_scratch(int M, int K, int N);
int t3d_pwconv_bwd_yfree(const void* dz, const void* x, const void* wd, const void* x_raw, const t3d_prologue* pro_in,
                         const void* residual, void* dx, double* stats, void* scratch, long long scratch_bytes, int M, int HW,
                         int K, int N, void* stream);
int t3d_pwconv_wgrad_yfree_finish(void* scratch, const t3d_bnbwd* bb, const void* w, float* dw, int M, int K, int N, void* stream);

/* Depthwise conv backward: data gradient and weight gradient in one pass.
 *   dz, y [B,Ho,Wo,C]: gradient at / raw input of the BatchNorm after the conv; bb its backward affine;
 *   w [C,k*k] fp32; x [B,H,W,C] + pro: tensor the forward conv read and how it was activated (no SE);
 *   residual [B,H,W,C] or NULL is added; dx [B,H,W,C];
 *   stats [2*C] fp64 or NULL: += sum(dx), sum(dx*x); dw [C,k*k] fp32 or NULL: += weight gradient. */
int t3d_dwconv_bwd(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const float* w,
                   const void* x, const t3d_prologue* pro, const void* residual, void* dx, double* stats,
                   float* dw, int B, int H, int W, int C, int k, int stride, void* stream);

/* BatchNorm backward bookkeeping: from the reductions the gradient-producing kernel emitted,
 *   stats [2*C] fp64 = sum(dz), sum(dz*y)  over `count` elements (dz: gradient at the BN output,
 *   y: raw BN input), and the forward's mean / invstd, build the backward affine
 *   dy = alpha*dz + beta*y + gammac  and the parameter gradients (autograd of nn.BatchNorm2d in
 *   training mode, mobilenetv3.py:113 etc.):
 *     dbeta = sum(dz), dgamma = invstd*(sum(dz*y) - mean*sum(dz)),
 *     alpha = gamma*invstd, beta = -alpha*invstd*dgamma/count, gammac = -alpha*sum(dz)/count - beta*mean.
 *   dgamma / dbeta (may be NULL) are OVERWRITTEN. */
int t3d_bn_bwd_finalize(const double* stats, int C, double count, const float* gamma, const float* mean,
                        const float* invstd, float* alpha, float* beta, float* gammac, float* dgamma,
                        float* dbeta, void* stream);

/* Stem patch gather: crops x [B,3,H,W] fp32 NCHW (the reference's input contract,
 * dataloaders/objectron_main.py:51-96) -> col [B*Ho*Wo, 32] (dtype), the 3x3 / stride-2 / pad-1
 * patches of nn.Conv2d(3,C,3,2,1) (mobilenetv3.py:110-115) in (ci,ky,kx) order, columns 27..31 zero.
 * The stem conv itself then runs as t3d_pwconv_{fwd,wgrad} with K = 32. */
int t3d_stem_im2col(int dtype, const float* x, void* col, int B, int H, int W, void* stream);
/* The same gather from raw uint8 NHWC crops [B,H,W,3], normalised on the way: (u/255 - mean[c]) * inv_std[c]
 * (configs/default_config.py:9-10; padding taps are zeros of the NORMALISED image, as the reference's conv sees them).
 * The crops can then stay uint8 from the decoder to the GPU: 4x less PCIe / HBM than the fp32 tensor. */
int t3d_stem_im2col_u8(int dtype, const unsigned char* x, const float* mean, const float* inv_std, void* col, int B, int H,
                       int W, void* stream);

/* A whole inverted-residual block in inference mode (running statistics: every BatchNorm is a per-channel affine), bf16,
 * for the 14x14 / 7x7 stages: z = BN3(W2 * act2(BN2(dw3x3(act1(BN1(W1 * x)))))) [+ x], models/mobilenetv3.py:146-164 in
 * eval().  One workgroup per image keeps the expanded tensors in LDS (csrc/block_eval.hip); replaces
 * t3d_pwconv_fwd + t3d_dwconv_fwd + t3d_pwconv_fwd + t3d_bn_apply of that block with the same rounding points.
 * x [B,H,W,Cin] finished input, w1 [Ce,Cin], w2 [Cout,Ce] bf16, wdw [Ce,9] fp32, scale/shift fp32 per channel (eval affines),
 * z [B,H,W,Cout].  Stride 1, 3x3, no squeeze-excite; shapes outside the built set return T3D_ERR_UNSUPPORTED. */
int t3d_ir_block_eval(const void* x, const void* w1, const float* scale1, const float* shift1, int act1, const float* wdw,
                      const float* scale2, const float* shift2, int act2, const void* w2, const float* scale3,
                      const float* shift3, int residual, void* z, int B, int H, int W, int Cin, int Ce, int Cout, void* stream);

/* Two-stage inference, input side: crop every detection out of ONE full frame and resize it to the regressor's input
 * size, in one launch.  frame [H,W,3] uint8 (device), rects [n,4] int32 (x0,y0,x1,y1), device) -> out [n,oh,ow,3] uint8
 * NHWC.  Replaces the host loop `frame[y0:y1, x0:x1]` + `cv.resize(crop, (w, h))` per detection (utils/ie_wrappers.py:
 * 18-21,128-133,154-158; same crop as dataloaders/objectron_main.py:98-127): numpy slice semantics (bounds clamped to
 * the frame; an empty crop gives zeros) and cv::resize INTER_LINEAR's 8-bit arithmetic (half-pixel centres, replicated
 * border, 11-bit fixed-point weights) as restated in csrc/crop.hip -- bit-exact against oracle/crop_resize.py. */
int t3d_crop_resize_u8(const unsigned char* frame, const int* rects, unsigned char* out, int n, int H, int W, int oh, int ow,
                       void* stream);

/* Materialise a block output:  z = act(scale*y + shift) + residual   (residual may be NULL; scale NULL = identity).
 * Replaces the BatchNorm normalise pass + `x + self.conv(x)` (mobilenetv3.py:159,162-164). y,z,residual [M,C]. */
int t3d_bn_apply(int dtype, const void* y, const t3d_prologue* pro, const void* residual, void* z, int M, int C,
                 void* stream);

/* Round 6: the BatchNorm statistics of an expansion conv from the Gram matrix of its narrow input (csrc/gram.hip).  For
 * y1 = W1 z (models/mobilenetv3.py:146-148: nn.Conv2d(inp, hidden_dim, 1) + nn.BatchNorm2d(hidden_dim)), sum(y1) = W1 (1^T z)
 * and sum(y1^2)_c = w_c^T (z^T z) w_c: one pass over the NARROW tensor replaces the statistics the 1x1 conv's epilogue takes
 * over the 6x wider one -- which is what lets the fused expand + depthwise forward (t3d_expdw_fwd) run in TRAINING mode.
 *   t3d_bn_apply_gram: t3d_bn_apply (z may be NULL: y is the finished tensor already, pro and residual NULL) that also adds
 *     [upper triangle of z^T z, row-major (i, j >= i) | 1^T z] into gram [16][K(K+1)/2 + K] fp64 -- 16 reduction replicas, a
 *     workgroup adds into one of them (caller zeroes all; order-independent adds); bf16, K in {8, 16}; the sums are those of the
 *     STORED (rounded) z.
 *   t3d_gram_bn_finalize: t3d_bn_finalize's outputs (scale, shift, mean, invstd, running statistics, num_batches_tracked) for
 *     the C channels of W1 z from those sums (the 16 replicas added first); w [C,K] in the conv's storage dtype (bf16: the matrix
 *     the MFMA multiplies). */
int t3d_bn_apply_gram(int dtype, const void* y, const t3d_prologue* pro, const void* residual, void* z, double* gram, int M,
                      int K, void* stream);
int t3d_gram_bn_finalize(const double* gram, const void* w, int C, int K, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                         float* scale, float* shift, float* mean_out, float* invstd_out, void* stream);

/* Backward of t3d_bn_apply's activation:  dzp = dz * act'(scale*y + shift);
 * stats [2*C] fp64 += sum(dzp), sum(dzp*y) (caller zeroes). */
int t3d_bn_act_bwd(int dtype, const void* dz, const void* y, const t3d_prologue* pro, void* dzp, double* stats,
                   int M, int C, void* stream);

/* Global average pool of the activated last feature map (model_builder.py:96-110, mode 'avg'):
 *   pooled[b][c] = mean_hw act(scale*y + shift),  y [B,HW,C] raw (dtype), pooled [B,C] fp32. */
int t3d_gap_fwd(int dtype, const void* y, const t3d_prologue* pro, float* pooled, int B, int HW, int C,
                void* stream);

/* ... and its backward: dz[b][hw][c] = dpooled[b][c]/HW * act'(scale*y + shift) (dtype),
 * stats [2*C] fp64 += sum(dz), sum(dz*y). */
int t3d_gap_bwd(int dtype, const float* dpooled, const void* y, const t3d_prologue* pro, void* dz, double* stats,
                int B, int HW, int C, void* stream);

/* The other pooling modes of ModelWrapper._glob_feature_vector (model_builder.py:96-110): 'max'
 * (F.adaptive_max_pool2d(x, 1)) and 'avg+max' (their sum).  argmax [B,C] int32 receives the (h*W + w) position of
 * each maximum -- the first one in scan order among equal values, which is where PyTorch's backward routes the
 * gradient -- and is what t3d_pool_bwd reads back (may be NULL for T3D_POOL_AVG). */
enum { T3D_POOL_AVG = 0, T3D_POOL_MAX = 1, T3D_POOL_AVGMAX = 2 };
int t3d_pool_fwd(int dtype, const void* y, const t3d_prologue* pro, int mode, float* pooled, int* argmax, int B,
                 int HW, int C, void* stream);
int t3d_pool_bwd(int dtype, const float* dpooled, const void* y, const t3d_prologue* pro, int mode, const int* argmax,
                 void* dz, double* stats, int B, int HW, int C, void* stream);

/* Regression + class heads (ModelWrapper.forward, model_builder.py:126-146):
 *   f' = act(scale*f + shift)                       (classifier BatchNorm1d + h_swish, or identity: pro NULL)
 *   kp[b]     = sigmoid(Wreg[cats[b]] f'[b] + breg[cats[b]])      [B,18]   (:137-139, class-gathered GEMV)
 *   logits[b] = Wcls (f'[b] * mask[b]) + bcls                     [B,ncls] (:142; mask = Dropout(0.5) keep/scale
 *               factors {0,2}, NULL in eval mode); logits may be NULL (num_classes == 1, :144).
 * f [B,F] fp32, Wreg [9,18,F], breg [9,18], Wcls [ncls,F], cats int64 [B]. */
int t3d_head_fwd(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* wreg,
                 const float* breg, const float* wcls, const float* bcls, const float* mask, float* kp,
                 float* logits, int B, int F, int ncls, void* stream);

/* Export-mode heads (ModelWrapper.forward_to_onnx, model_builder.py:112-124): EVERY one of the 9 regressors applied
 * to every sample in one launch, kp_all [9,B,18] = sigmoid(Wreg[k] f'[b] + breg[k]); logits as above (eval: no mask). */
int t3d_head_fwd_all(const float* f, const t3d_prologue* pro, const float* wreg, const float* breg, const float* wcls,
                     const float* bcls, float* kp_all, float* logits, int B, int F, int ncls, void* stream);

/* A single head applied on its own -- `model.regressors[k](f)` / `model.cls_fc[1](f)` for callers that use the
 * reference's attributes directly (model_builder.py:79-85): y [M,N] = x [M,K] w[N,K]^T + bias, fp32, any N. */
int t3d_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, void* stream);

/* Head backward.  dkp [B,18], dlogits [B,ncls] (may be NULL) -> df [B,F] (when pro is given: the gradient
 * at the BatchNorm1d OUTPUT, i.e. already multiplied by act'; then stats [2*F] fp64 += sum(df), sum(df*f)),
 * dwreg [9,18,F], dbreg [9,18], dwcls [ncls,F], dbcls [ncls] are ACCUMULATED (+=, batch slices in parallel): the
 * caller zeroes them once per step, like every other weight-gradient buffer of this ABI.
 * dpre [B,18] fp32 scratch. */
int t3d_head_bwd(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* wreg,
                 const float* wcls, const float* mask, const float* kp, const float* dkp, const float* dlogits,
                 float* dpre, float* df, double* stats, float* dwreg, float* dbreg, float* dwcls, float* dbcls,
                 int B, int F, int ncls, void* stream);
/* The weight / bias gradients of t3d_head_bwd on their own: pass dwreg = NULL there (data gradient only, dpre is left
 * behind) and issue this on the weight-gradient stream -- 44 us off the start of the backward's critical chain. */
int t3d_head_bwd_weights(const float* f, const t3d_prologue* pro, const int64_t* cats, const float* mask, const float* dpre,
                         const float* dlogits, float* dwreg, float* dbreg, float* dwcls, float* dbcls, int B, int F, int ncls,
                         void* stream);

/* Keypoint / class losses of torchdet3d/losses/regression_losses.py + builders/loss_builder.py:7-28,
 * combined as LossManager.parse_losses does (:79-95), value AND gradient in one launch, plus the
 * training metrics of evaluation/metrics.py:10-37.  A coefficient of 0 disables a term.
 *   total = lam_reg * sum_i c_i * L_i(kp, gt)  +  lam_cls * c_ce * CE(logits, cats)
 * out[0] total, out[1] sum_i c_i L_i, out[2] c_ce*CE, out[3] ADD (mean), out[4] SADD (mean), out[5] accuracy,
 * out[6] sum ADD / 9 ... (metrics with reduce_mean=False: out[6] ADD, out[7] SADD, out[8] correct count).
 * dkp [B,18] / dlogits [B,ncls] (may be NULL: values only) receive d total / d input. */
typedef struct {
  float c_l1, c_mse, c_smoothl1, smoothl1_beta, c_add, c_diag, c_wing, wing_w, wing_eps, c_ce;
  float lam_reg, lam_cls;
} t3d_loss_cfg;
int t3d_loss_fwd_bwd(const t3d_loss_cfg* cfg, const float* kp, const float* gt_kp, const float* logits,
                     const int64_t* cats, float* out, float* dkp, float* dlogits, int B, int ncls, void* stream);

/* Per-sample summands of the validation metrics (torchdet3d/evaluation/metrics.py:39-68 `compute_metrics_per_cls`, which
 * calls compute_average_distance / compute_accuracy with reduce_mean=False on every class subset, :48-55):
 * out [B][3] = { sum_k ||p_k - t_k|| / 9, symmetric-ADD summand / 9 (:13-21), arg-max hit (:31-37; logits NULL: the
 * width-1 "targets" of num_classes == 1, arg-max always 0) } -- the host adds them per class: one launch and one
 * read-back per validation batch instead of two launches and two syncs per class present. */
int t3d_metrics_per_sample(const float* kp, const float* gt_kp, const float* logits, const int64_t* cats, float* out,
                           int B, int ncls, void* stream);

/* 2-D based 3-D IoU of the validation loop, batched on the device (torchdet3d/evaluation/metrics.py:70-89 with
 * torchdet3d/utils/geometry.py:51-108 `lift_2d(..., portrait)` and the objectron box fit + box-box IoU it calls):
 *   pred_kp, gt_kp [B,9,2] fp32 normalised keypoints (the centre keypoint 0 is not used by the lift, :71-72);
 *   camera_ndc: NULL for the default camera (geometry.py:16-19,29-37: fx = fy = 2, cx = cy = 0) or {fx, fy, cx, cy} fp64 (HOST);
 *   iou [B] fp64 per-sample IoU (0 where the reference swallows a QhullError / LinAlgError, metrics.py:82-86), may be
 *   NULL when only `lifted` is wanted; total: fp64 scalar += sum(iou) (may be NULL; caller zeroes);
 *   lifted [B,2,9,3] fp64 or NULL: the lifted vertices of (pred, gt) -- what lift_2d returns. */
int t3d_iou3d(const float* pred_kp, const float* gt_kp, int B, int portrait, const double* camera_ndc, double* iou,
              double* total, double* lifted, void* stream);

/* The box-box part alone: verts [B,2,9,3] fp64 = B pairs of 9-vertex boxes in objectron's vertex order (centre, then
 * the 8 corners), as `box.Box(vertices)` takes them (metrics.py:79-81) -> iou [B] fp64. */
int t3d_box_iou3d(const double* verts, int B, double* iou, double* total, void* stream);

/* Squeeze-excite gate (SELayer, mobilenetv3.py:92-107) from the depthwise kernel's per-sample sums:
 *   m = scale*gap_sum/HW + shift (= mean_hw of the BatchNorm output), h = relu(W1 m + b1), q = W2 h + b2,
 *   s = h_sigmoid(q).  gap_sum, m, q, s [B,C]; h [B,R]; W1 [R,C]; W2 [C,R]; all fp32. */
int t3d_se_fwd(const float* gap_sum, const float* scale, const float* shift, const float* w1, const float* b1,
               const float* w2, const float* b2, float* m, float* h, float* q, float* s, int B, int C, int R, int HW,
               void* stream);

/* Squeeze-excite applied AFTER the activation (the no-expand block layout, mobilenetv3.py:138-140: dw -> BN -> act ->
 * SE -> 1x1; MobileNetV3-small features.1): v = s*a, a = act(scale*y + shift).  Forward: t3d_gap_fwd gives mean_hw(a),
 * t3d_se_fwd (scale = 1, shift = 0, HW = 1) the gate.  Backward, with dv [B*HW,C] the gradient at the gated tensor:
 *   _sums : ps [B,C,2] = { sum_hw dv*a, 0 }   -> t3d_se_bwd with scale = 0, shift = 1 (so that ds = ps[..][0]) gives g
 *   _apply: du = (s*dv + g) * act'(scale*y + shift) at the BatchNorm output, stats [2*C] fp64 += sum(du), sum(du*y). */
int t3d_se_after_sums(int dtype, const void* dv, const void* y, const t3d_prologue* pro, float* ps, int B, int HW, int C,
                      void* stream);
int t3d_se_after_apply(int dtype, const void* dv, const void* y, const t3d_prologue* pro, const float* s, const float* g,
                       void* du, double* stats, int B, int HW, int C, void* stream);

/* Backward of the gate.  ps_stats [B,C,2] = per-sample sum_hw(dv), sum_hw(dv*y) from t3d_pwconv_dgrad
 * (dv: gradient at the gated tensor, y: raw depthwise output).  Produces g [B,C] (the pooled path's
 * per-pixel gradient, so that du = s*dv + g), accumulates the depthwise BatchNorm's backward sums
 * stats [2*C] fp64 += sum(du), sum(du*y), and OVERWRITES the FC gradients dw1 [R,C], db1 [R], dw2 [C,R], db2 [C].
 * dq [B,C], dp [B,R]: scratch. */
int t3d_se_bwd(const float* ps_stats, const float* gap_sum, const float* scale, const float* shift, const float* w1,
               const float* w2, const float* m, const float* h, const float* q, const float* s, float* g, float* dq,
               float* dp, double* stats, float* dw1, float* db1, float* dw2, float* db2, int B, int C, int R, int HW,
               void* stream);

/* The same gate with ONE launch per direction (a workgroup per 8 samples walks both FCs; csrc/se.hip): what the host side
 * uses.  _fwd_fused takes TRANSPOSED fp32 copies of the FC weights (w1t [C][R] = fc.0.weight^T, w2t [R][C] = fc.2.weight^T;
 * e.g. kept current by t3d_pack_weights_batched(T3D_F32, ...)) and otherwise t3d_se_fwd's arguments.  _bwd_data is
 * t3d_se_bwd without the weight gradients (dq, dp stay behind for them; stats may be NULL, and is spread over the
 * reduction replicas of t3d_set_reduction_replicas like the conv kernels' sums); _bwd_weights computes dw1, db1, dw2, db2 from
 * m, h, dq, dp -- leaves of the backward graph, issued on the weight-gradient stream. */
int t3d_se_fwd_fused(const float* gap_sum, const float* scale, const float* shift, const float* w1t, const float* b1,
                     const float* w2t, const float* b2, float* m, float* h, float* q, float* s, int B, int C, int R, int HW,
                     void* stream);
int t3d_se_bwd_data(const float* ps_stats, const float* gap_sum, const float* scale, const float* shift, const float* w1,
                    const float* w2, const float* h, const float* q, const float* s, float* g, float* dq, float* dp,
                    double* stats, int B, int C, int R, int HW, void* stream);
int t3d_se_bwd_weights(const float* m, const float* h, const float* dq, const float* dp, float* dw1, float* db1, float* dw2,
                       float* db2, int B, int C, int R, void* stream);

/* t3d_bn_apply + t3d_pwconv_fwd in one launch: the 1x1 conv whose input is the previous block's OUTPUT, still in its raw
 * form (models/mobilenetv3.py:158-166: `x + conv(x)` feeding the next block's expansion).  The operand
 *   z = round_to_storage(act(scale*y_in + shift) + residual)
 * is formed on load exactly as t3d_bn_apply would have stored it, used for the contraction, and written to z_out once
 * (the skip connection and the backward need it).  bf16 streaming kernel; other cases run the two launches.
 *   y_in [M,K] raw, pro_in: its BatchNorm affine (+ activation), residual [M,K] or NULL, z_out [M,K],
 *   w [N,K], y [M,N] raw output, stats as in t3d_pwconv_fwd. */
int t3d_pwconv_fwd_mat(int dtype, const void* y_in, const t3d_prologue* pro_in, const void* residual, void* z_out,
                       const void* w, void* y, double* stats, int M, int HW, int K, int N, void* stream);

/* SSD detector post-processing (configs/detection/mnv2_ssd_300_2_heads.py:15-39,65-69: SSDHead outputs of <= 2 feature
 * maps -> DeltaXYWH decode -> softmax -> per-class greedy NMS), one launch per frame batch.  The implementing mmdetection
 * fork is external to the reference (README.md:56-57): arithmetic per the published mmdet definitions, parity unpinned.
 *   cls[l] [B*hw[l]][cls_stride[l]]: logits, channel = anchor*(num_classes+1) + class, background LAST;
 *   reg[l] [B*hw[l]][reg_stride[l]]: deltas, channel = anchor*4 + (dx, dy, dw, dh); storage dtype;
 *   anchors [sum hw*nanchors][4] fp32 (x1, y1, x2, y2 in input pixels), level-major, then pixel, then anchor;
 *   stds [4]; boxes are clipped to (img_w, img_h).
 *   out [B][num_classes][max_per_class][6] = x1, y1, x2, y2, score, label in NMS order; counts [B][num_classes].
 * cls / reg / hw / nanchors / *_stride are HOST arrays of length nlevels. */
int t3d_ssd_decode_nms(int dtype, int nlevels, const void* const* cls, const void* const* reg, const int* hw,
                       const int* nanchors, const int* cls_stride, const int* reg_stride, const float* anchors, int B,
                       int num_classes, float score_thr, float iou_thr, int max_per_class, float img_w, float img_h,
                       const float* stds, float* out, int* counts, void* stream);

/* ---- ResNet-50 backbone (BASELINE config 4; the reference has no ResNet -- standard torchvision architecture, parity
 * against oracle/resnet.py, unpinned).  Dense k x k convolutions run as patch gather + the pointwise GEMM entry points
 * (t3d_pwconv_fwd / _dgrad / _wgrad with K = Kp); everything below is an HBM-bound gather / elementwise kernel. ---- */

/* nn.Conv2d(C, N, k, stride, pad) input side: x [B,H,W,C] raw tensor read through `pro` (BatchNorm affine + activation of
 * its producer; NULL: finished tensor), zero padding applied to the ACTIVATED tensor -> col [B*Ho*Wo][Kp],
 * column (ky*k + kx)*C + c, columns k*k*C .. Kp-1 zero (Kp: multiple of 8).  _nchw: the fp32 NCHW crops of the input
 * contract (stem). */
int t3d_im2col(int dtype, const void* x, const t3d_prologue* pro, void* col, int B, int H, int W, int C, int k, int stride,
               int pad, int Kp, void* stream);
int t3d_im2col_nchw(int dtype, const float* x, void* col, int B, int H, int W, int C, int k, int stride, int pad, int Kp,
                    void* stream);
/* Backward of t3d_im2col: dcol [B*Ho*Wo][Kp] (= dy W, from t3d_pwconv_dgrad) -> dx [B,H,W,C] = gradient at the producer's
 * BatchNorm output (patch gradients summed per input pixel, times act'(scale*x_raw + shift)); stats [2*C] fp64 or NULL:
 * += sum(dx), sum(dx * x_raw) (replicas as elsewhere). */
int t3d_col2im_bwd(int dtype, const void* dcol, const void* x_raw, const t3d_prologue* pro, void* dx, double* stats, int B,
                   int H, int W, int C, int k, int stride, int pad, int Kp, void* stream);
/* [N][C][k][k] fp32 master weight -> [N][Kp] GEMM operand in patch-column order (storage dtype), and the gradient back. */
int t3d_pack_conv_weight(int dtype, const float* w, void* out, int N, int C, int k, int Kp, void* stream);
int t3d_unpack_conv_grad(const float* dw_packed, float* dw, int N, int C, int k, int Kp, void* stream);
/* nn.MaxPool2d(3, 2, 1) over act(scale*y + shift): out [B,Ho,Wo,C] finished tensor, argmax [B,Ho,Wo,C] uint8 (window
 * position 0..8, first maximum in scan order); backward: dy = gradient at y's BatchNorm output + its backward sums. */
int t3d_maxpool_fwd(int dtype, const void* y, const t3d_prologue* pro, void* out, unsigned char* argmax, int B, int H, int W,
                    int C, void* stream);
int t3d_maxpool_bwd(int dtype, const void* dout, const unsigned char* argmax, const void* y, const t3d_prologue* pro, void* dy,
                    double* stats, int B, int H, int W, int C, void* stream);
/* Bottleneck tail  z = relu(BN3(y3) + shortcut):  shortcut read through pro_s (projection shortcut: raw 1x1 output +
 * its BatchNorm) or as it is (identity, pro_s NULL).  Backward: g = dz * [z > 0] (the gradient at BOTH BatchNorm
 * outputs); stats3 += sum g, sum g*y3; statsd (with yd) += sum g, sum g*yd. */
int t3d_res_relu_fwd(int dtype, const void* y3, const t3d_prologue* pro3, const void* shortcut, const t3d_prologue* pro_s,
                     void* z, int M, int C, void* stream);
int t3d_res_relu_bwd(int dtype, const void* dz, const void* z, const void* y3, const void* yd, void* g, double* stats3,
                     double* statsd, int M, int C, void* stream);
/* upsample == 0: out [B,Ho,Wo,C] = x[:, ::stride, ::stride, :] (x [B,H,W,C]);  upsample == 1: out [B,H,W,C] = x [B,Ho,Wo,C]
 * at the multiples of stride, zero elsewhere (gradient of the former). */
int t3d_subsample(int dtype, const void* x, void* out, int B, int H, int W, int C, int stride, int upsample, void* stream);

/* Reduction replicas.  Every `+=` reduction output of the kernels (the fp64 BatchNorm sums `stats`, the depthwise
 * weight gradient `dw`) is hit by one atomic per channel per workgroup; with hundreds of workgroups on a few KB of
 * addresses those atomics serialise.  With nrep > 1 the streaming kernels add into replica (workgroup % nrep):
 *   stats replica r lives at  stats + r*stats_stride  (doubles);   dw replica r at  dw + r*C*k*k  (floats);
 * t3d_bn_finalize / t3d_bn_bwd_finalize sum the nrep stats replicas, the caller sums the dw replicas.
 * Process-wide setting (default 1 = plain behaviour); kernels that do not implement replicas use replica 0. */
int t3d_set_reduction_replicas(int nrep, long long stats_stride);

/* Depthwise weight gradient without atomics (bit-reproducible).  With capacity > 0 the NEXT t3d_dwconv_bwd launches treat
 * `dw` as [capacity][C][k*k] fp32 SLOTS: every workgroup of a streaming kernel stores (does not add) its partial weight
 * gradient into its own slot and *used_out (device memory) receives the number of slots the launch filled; the caller adds
 * slots 0 .. *used_out-1 in index order (t3d_sum_slots_batched), which no longer depends on the arrival order of ~500
 * workgroups (torch's autograd makes no such promise for models/mobilenetv3.py:151 either; this is about run-to-run
 * reproducibility of the training step).  A launch that needs more workgroups than `capacity`, and the LDS-tiled fallback
 * kernel, add atomically into the first nrep slots (t3d_set_reduction_replicas) and report *used_out = nrep: those slots
 * must be zero before the launch.  capacity == 0 (default): the replica behaviour above.  Process-wide setting. */
int t3d_set_dw_slots(int capacity, int* used_out);

/* Squeeze-excite pooled sums without order noise.  on != 0: every `pooled` / `gap_sum` argument of the NEXT t3d_dwconv_fwd,
 * t3d_se_fwd(_fused), t3d_se_bwd(_data) calls addresses int64 [B][C] in units of 2^-24 instead of float [B][C] (zero it
 * before the depthwise launch, as before): the depthwise kernel's work items add integers, which is associative, so the
 * sums -- and the gate torch computes from `y.mean((2, 3))` at models/mobilenetv3.py:100-104 -- are bit-reproducible from
 * run to run.  on == 0 (default): fp32 atomics into float [B][C].  Process-wide setting. */
int t3d_set_exact_pool(int on);

/* Measurement aid (bench.py's roofline block): attach two hipEvent_t (created with timing enabled, recorded at least once) to
 * the NEXT depthwise-convolution kernel launch -- t3d_dwconv_fwd / t3d_dwconv_bwd -- so that hipEventElapsedTime(start, stop)
 * is that kernel's own begin-to-end time (hipExtLaunchKernelGGL; what rocprofv3 --kernel-trace reports).  The pair is
 * consumed by the launch; NULL, NULL clears it. */
int t3d_set_launch_events(void* start_event, void* stop_event);
/* desc = n rows of int64 {src [slots][count] fp32, dst [count] fp32, count, used (device int*)}:
 * dst = src[0] + src[1] + ... + src[*used - 1], in that order (overwrites dst). */
int t3d_sum_slots_batched(const long long* desc, int n, void* stream);

/* BatchNorm finalize derived by the CONSUMER.  t3d_bn_finalize / t3d_bn_bwd_finalize are 5-us launches that sit
 * between every convolution and its consumer (~100 per training step, all on the critical stream: 0.64 ms of an
 * 8.4-ms MobileNetV2 step with the launch gaps).  A fold request names a coefficient array (`key`: the `scale` pointer
 * of a t3d_prologue, or the `alpha` pointer of a t3d_bnbwd) and a DEVICE-resident descriptor of everything the finalize
 * would compute; the NEXT launch that takes exactly that array as its prologue / BatchNorm-backward coefficients
 * consumes the request: its workgroups derive the coefficients of their own channels from the replica sums (the
 * kernel boundary already orders the sums before them -- no device-wide barrier), and one workgroup per channel also
 * writes the finalize's outputs for every later reader.  Implemented by the bf16 streaming kernels (t3d_pwconv_fwd /
 * _dgrad / _wgrad (the weight gradient derives without publishing), t3d_dwconv_fwd / _bwd with k = 3) and
 * t3d_bn_apply, t3d_se_bwd_affine (round 6); other launches
 * leave the request pending: check t3d_fold_pending() (returns 1 and clears it) right after the launch -- the launch
 * then read unfinalized coefficients and must be treated as failed (the entry points listed above fall back to a
 * finalize launch of their own on code paths without the derive prologue, e.g. fp32 storage).
 *   kind 1 (forward, = t3d_bn_finalize):      stats = sum(y), sum(y^2); o0..o3 = scale, shift, mean, invstd
 *   kind 2 (backward, = t3d_bn_bwd_finalize): stats = sum(dz), sum(dz*y); o0..o4 = alpha, beta, gammac, dgamma, dbeta */
typedef struct {
  int kind;
  int C;
  int nrep;                  /* reduction replicas behind `stats` ... */
  long long rstride;         /* ... replica r at stats + r*rstride (doubles) */
  const double* stats;
  double count;
  const float *gamma, *beta;
  float *rm, *rv;            /* kind 1: running estimates (may be NULL) */
  long long* nbt;            /* kind 1: num_batches_tracked (may be NULL) */
  float momentum, eps;
  float *o0, *o1, *o2, *o3, *o4;
  const float *mean, *invstd; /* kind 2: the forward's batch mean / invstd */
} t3d_bn_fold;
int t3d_fold_request(const t3d_bn_fold* desc_device, const void* key);
int t3d_fold_pending(void);

/* Optional device scratch (caller-owned, process-wide setting; NULL/0 clears it).  With a workspace the bf16
 * pointwise weight gradient writes its per-split partial dW tiles there with plain stores and reduces them in a
 * second, deterministic pass; without one the partials leave as fp32 atomics.  64 MB covers every layer shape of
 * the supported models at batch 256. */
int t3d_set_workspace(void* ptr, long long bytes);
/* Scratch for launches on the caller's MAIN stream (the workspace above belongs to the weight-gradient launches, which the
 * host side issues on a second stream): the fp32 pointwise kernel splits the contraction of few-pixel, deep layers over it
 * (classifier Linear(960, 1280) on 256 samples: 16 workgroups x 40 serial rounds otherwise).  NULL / 0 = unsplit. */
int t3d_set_main_workspace(void* ptr, long long bytes);

/* All weight matrices of a model in ONE launch: desc is a DEVICE array of n records of 7 int64
 * {src fp32 [rows,cols], out [rows,cols] or 0, out_t [cols,rows] or 0, rows, cols, frag or 0, frag_t or 0} (outputs in `dtype`);
 * frag / frag_t: fragment-order copies of the matrix / of its transpose (t3d_pwconv_pack_frag; t3d_pwconv_frag_bytes(rows, cols)
 * / (cols, rows) bytes, cleared once by the caller: only the elements of the matrix are written). */
int t3d_pack_weights_batched(int dtype, const long long* desc, int n, void* stream);

/* Optimizer step: torch.optim.AdamW as build_optimizer(name='adam') constructs it
 * (torchdet3d/builders/optim_builder.py:10-12; stepped at trainer/train.py:50-52) over ONE flat fp32 buffer of n
 * elements (n % 4 == 0): decoupled weight decay, no amsgrad.  `step` is the 1-based step count (bias corrections are
 * computed on the host in fp64), g is read as grad_scale * g (1/world for summed data-parallel gradients), m / v are the
 * caller-owned moment buffers (zeroed before the first step). */
int t3d_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, long long step, double grad_scale, void* stream);
/* Divergence watch for the optimizer steps that follow (process-wide, like t3d_set_reduction_replicas): with a DEVICE
 * int64 word set (the caller initialises it to INT64_MAX), t3d_adamw_step does *word = min(*word, step) when it meets a
 * non-finite gradient element; NULL switches it off.  Serves the loss the reference's loop reads at train.py:57 -- after a
 * divergence `loss.item()` is NaN there, while the clamp-form ReLU6 of the 16-bit kernels can leave this path's loss
 * finite: the trainer reports NaN from the first watched step on, per step and identically on every rank (the gradient is
 * all-reduced before the optimizer reads it). */
int t3d_set_grad_watch(long long* first_bad_step);

/* Small bookkeeping kernels that keep the step free of framework arithmetic:
 *  _copy_cols   dst [rows,cols_dst] <- leading columns of src [rows,cols_src], rest zero (the stem's [C,27] <-> [C,32]
 *               patch-row weights and their gradient, mobilenetv3.py:110-115);
 *  _bn_bias_grad gradient of the bias of a Linear that feeds a train-mode BatchNorm1d (classifier, mobilenetv3.py:191-194):
 *               sum_b dy = alpha*sum(dz) + beta*sum(y) + count*gammac from the forward / backward sum replicas;
 *  _se_bwd_affine the per-sample BatchNorm-backward affine behind a squeeze-excite gate: aps = s*alpha, gps = gammac + g*alpha
 *               ([B,C]; what t3d_bnbwd.per_sample reads); serves a pending t3d_fold_request for `alpha` (derives AND publishes
 *               alpha / beta / gamma + the BatchNorm's parameter gradients, one owner workgroup per 32 channels);
 *  _dropout_mask nn.Dropout(p) keep/scale factors {0, 1/(1-p)} (model_builder.py:83) from Philox-4x32-10 keyed by
 *               (seed, offset): stateless and reproducible. */
int t3d_copy_cols(const float* src, float* dst, int rows, int cols_src, int cols_dst, void* stream);
int t3d_bn_bias_grad(const double* fwd_stats, const double* bwd_stats, int C, double count, const float* alpha,
                     const float* beta, const float* gammac, float* dbias, void* stream);
int t3d_se_bwd_affine(const float* s, const float* g, const float* alpha, const float* gammac, float* aps, float* gps,
                      int B, int C, void* stream);
int t3d_dropout_mask(float* mask, long long n, unsigned long long seed, unsigned long long offset, float p, void* stream);

/* Zero fill of several device buffers in one launch: desc = n rows of int64 {ptr, bytes}, bytes % 16 == 0, DEVICE array
 * (the per-step clears of the gradient buffer, the BatchNorm sum replicas and the depthwise weight-gradient replicas). */
int t3d_zero_batched(const long long* desc, int n, void* stream);

/* Dense 3x3 convolution (pad 1, stride 1 or 2) as an implicit GEMM, bf16 storage (round 5; csrc/conv3x3.hip): the three GEMMs of
 * ResNet-50's conv2 layers gather their operand rows from the activation tensor themselves -- no patch matrix in HBM.  Replaces
 * nn.Conv2d(C, N, 3, stride, 1, bias=False) and its autograd in a torchvision Bottleneck, the backbone BASELINE config 4 builds
 * through torchdet3d/builders/model_builder.py:73-151 (t3d_im2col / t3d_col2im_bwd + the 1x1 kernels remain for fp32 storage
 * and the 7x7 stem).  C a power of two >= 32 (the weight gradient: N >= 64 as well), N % 8 == 0.
 *   _fwd    dtype = T3D_BF16 | T3D_W_FRAG; x [B,H,W,C] raw + `pro` (BatchNorm affine + activation of its producer, may be NULL);
 *           w_frag = t3d_pwconv_pack_frag of the [N][9C] patch-column-order weights (t3d_pack_conv_weight);
 *           y [B,Ho,Wo,N] raw, stats [2N] fp64 replicas or NULL as t3d_pwconv_fwd;
 *   _dgrad  dx [B,H,W,C] = gradient at the producer's BatchNorm output: sum over (tap, n) of the BatchNorm-backward affine of
 *           (dz, y) [B,Ho,Wo,N] times wd, times act'(x_raw through pro_in), stats += sum(dx), sum(dx * x_raw) as t3d_pwconv_dgrad;
 *           wd_frag = t3d_pwconv_pack_frag of the [C][9N] matrix t3d_pack_conv3x3_dgrad_weight writes (wd[c][t*N+n] = w[n][c][t]);
 *   _wgrad  dw_packed [N][9C] fp32 = (BatchNorm-backward affine of (dz, y))^T * gathered act(x) (written, not added to), patch-column order
 *           (t3d_unpack_conv_grad -> [N][C][3][3]); dtype T3D_BF16; the caller's workspace as t3d_pwconv_wgrad. */
int t3d_conv3x3_fwd(int dtype, const void* x, const t3d_prologue* pro, const void* w_frag, void* y, double* stats, int B, int H,
                    int W, int C, int N, int stride, void* stream);
int t3d_conv3x3_dgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* wd_frag, const void* x_raw,
                      const t3d_prologue* pro_in, void* dx, double* stats, int B, int H, int W, int C, int N, int stride,
                      void* stream);
int t3d_conv3x3_wgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* x, const t3d_prologue* pro,
                      float* dw_packed, int B, int H, int W, int C, int N, int stride, void* stream);
int t3d_pack_conv3x3_dgrad_weight(const float* w, void* out, int N, int C, void* stream);

/* Fused expand 1x1 conv + BatchNorm + activation + depthwise 3x3 conv forward of an inverted-residual block (round 5;
 * csrc/expdw_fwd.hip): y2 = dwconv3x3(act(scale1 * round(W1 z) + shift1)), dtype T3D_BF16 or T3D_F16 (inference), K <= 32 (the 112x112 .. 28x28 blocks of
 * MobileNetV2), act in {T3D_ACT_RELU, T3D_ACT_RELU6}.  Replaces nn.Conv2d(K, C, 1) + nn.BatchNorm2d(C) + activation +
 * nn.Conv2d(C, C, 3, s, 1, groups=C) + the statistics pass of the following BatchNorm2d (models/mobilenetv3.py:146-153) without
 * the expanded tensor's round trip through HBM: the stencil reads it out of LDS.
 *   z [B,H,W,K] finished block input, w1 [C,K] bf16, scale1 / shift1 [C] fp32: the expansion's BatchNorm affine -- its batch
 *   statistics must exist beforehand (a statistics-only pass: t3d_pwconv_fwd / _fwd_mat with y = NULL), wdw [C,9] fp32,
 *   y1 [B,H,W,C] raw expansion or NULL (stored only for a backward that reads it), y2 [B,Ho,Wo,C] raw depthwise output,
 *   stats2 [2*C] fp64 replicas or NULL: += sum(y2), sum(y2^2) (order-independent, as t3d_dwconv_fwd).
 * T3D_ERR_UNSUPPORTED for shapes it does not take (the caller runs t3d_pwconv_fwd + t3d_dwconv_fwd);
 * t3d_expdw_supported answers that question ahead of the launch: 1 when t3d_expdw_fwd would take (dtype, act, shape), else 0
 * (an input row wider than the workgroup's fragment registers / LDS rows hold -- W > 213 -- is the case the channel tests do not
 * show: 448 ... 512-pixel inputs reach MobileNetV2's second block at W = 224 ... 256). */
int t3d_expdw_fwd(int dtype, const void* z, const void* w1, const float* scale1, const float* shift1, int act, const float* wdw,
                  void* y1, void* y2, double* stats2, int B, int H, int W, int K, int C, int stride, void* stream);
int t3d_expdw_supported(int dtype, int act, int B, int H, int W, int K, int C, int stride);

/* Step plans (round 5; csrc/plan.hip): the whole train iteration as ONE host call.
 * Replaces the per-launch host loop of torchdet3d/trainer/train.py:44-55 + builders/optim_builder.py:10-12 (model forward,
 * losses, loss.backward(), optimizer.step()): ~230 enqueue-only calls of this header that are the same from step to step
 * except for a few values.  A plan is a recorded list of
 *   calls        any enqueue entry point of this header by NAME with its arguments as 64-bit words (pointers and integers
 *                as they are, float / double as their bit patterns), the three by-pointer structs (t3d_prologue, t3d_bnbwd,
 *                t3d_loss_cfg) COPIED into the plan;
 *   forks        hipEventRecord on one stream + hipStreamWaitEvent on another (the weight-gradient stream's dependencies);
 *   read-backs   an asynchronous device-to-host copy and an event record (the step's metrics on their way to the host),
 * and t3d_plan_run replays a segment of it through the same exported entry points (same host-side state, same launches:
 * bit-identical to issuing the calls one by one).  What changes per step goes through SLOTS: an argument recorded with
 * kind 2 takes its value from slots[index] at run time (input pointers of the batch, dropout counter, optimizer step
 * count, learning rate bits, pinned read-back address, event handle).
 *   kinds[i]: 0 literal word, 1 struct (words[i] = HOST address of the struct, struct_bytes[i] its size; copied), 2 slot
 *             (words[i] = slot index);
 *   _end_segment closes the current segment and RETURNS its index (segments let a caller interleave work of its own, e.g.
 *             an RCCL all-reduce of a gradient bucket, between two runs); _num_ops RETURNS the op count (kind -1: all, 0 calls,
 *             1 forks, 2 copies, 3 event records);
 *   _time_entry switches kernel-exact event timing (t3d_set_launch_events) on / off for one entry point; _run then attaches
 *             consecutive pairs of `events` (hipEvent_t, timing enabled) to that entry point's launches in op order;
 *   _run      segment >= 0, or -1 for the whole plan; RETURNS the number of events consumed (>= 0) or T3D_ERR_*;
 *             _failed_op RETURNS the index of the op that failed (-1: none) and its return code. */
typedef struct t3d_plan t3d_plan;
int t3d_plan_create(t3d_plan** out);
int t3d_plan_destroy(t3d_plan* plan);
int t3d_plan_add_call(t3d_plan* plan, const char* entry, int nargs, const int* kinds, const unsigned long long* words,
                      const int* struct_bytes);
int t3d_plan_add_fork(t3d_plan* plan, void* from_stream, void* to_stream);
/* The fork without a packet in the producing queue: `to_stream` waits for the last kernel that call op `producer_op` (index in
 * the plan, calls / forks / copies counted alike) launches on `from_stream`, through the STOP event of that kernel's own
 * dispatch (hipExtLaunchKernelGGL) -- a device-side hand-off: the main queue of the step, whose every microsecond is on the
 * critical path, carries no event records for the ~35 launches it hands to the weight-gradient stream.
 * t3d_launch_count: how many kernels the library has launched so far and on which stream the last one went (what a recorder
 * needs to know which call was the producer). */
int t3d_plan_add_fork_after(t3d_plan* plan, int producer_op, void* from_stream, void* to_stream);
int t3d_launch_count(unsigned long long* count, void** last_stream);
int t3d_plan_add_copy_d2h(t3d_plan* plan, int dst_slot, const void* src, long long bytes, void* stream);
int t3d_plan_add_event_record(t3d_plan* plan, int event_slot, void* stream);
int t3d_plan_end_segment(t3d_plan* plan);
int t3d_plan_num_ops(const t3d_plan* plan, int kind);
int t3d_plan_time_entry(t3d_plan* plan, const char* entry, int on);
int t3d_plan_failed_op(const t3d_plan* plan, int* rc_out);
int t3d_plan_run(t3d_plan* plan, int segment, const unsigned long long* slots, int nslots, void** events, int nevents);

#ifdef __cplusplus
}
#endif
#endif /* T3D_H_ */
