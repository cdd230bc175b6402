/*
 * edet_hip.h -- C ABI of the MI355X (gfx950) EfficientDet hot path.
 *
 * The reference (google/automl, efficientdet/) has no FFI boundary of its own:
 * its hot path is Python calling un-vendored TensorFlow ops.  Each entry point
 * below therefore replaces one *TensorFlow op call site* of the reference and
 * cites it (paths relative to the reference root).  The caller is the Python
 * host mirror in automl_amd/ (ctypes); see INTEGRATION.md for the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure;
 *     edet_last_error() returns a thread-local message.
 *   - all tensors are NHWC in device memory (HBM), element type `dtype`
 *     (EDET_F32 or EDET_BF16), channel stride `ld` >= c (ld % 8 == 0);
 *     statistics, gates, scales, parameters' master copies and parameter
 *     gradients are always fp32.
 *   - the caller owns every buffer; the library allocates nothing and only
 *     enqueues work on the HIP stream passed in (`stream` is a hipStream_t
 *     passed as void*; NULL = default stream).  Nothing synchronises, so all
 *     entry points are legal inside hipGraph stream capture.
 *   - not thread-safe per stream; re-entrant across streams.
 */
#ifndef EDET_HIP_H_
#define EDET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDET_F32 0
#define EDET_BF16 1

#define EDET_ACT_NONE 0
#define EDET_ACT_SWISH 1
/* utils.activation_fn (utils.py:36-53) beyond swish: value and derivative at every activated view */
#define EDET_ACT_RELU 2
#define EDET_ACT_RELU6 3
#define EDET_ACT_HSWISH 4   /* x * relu6(x + 3) / 6 */
#define EDET_ACT_MISH 5     /* x * tanh(softplus(x)) */
#define EDET_ACT_SRELU 6    /* utils.srelu_fn (utils.py:25-31): x - log(beta x + 1) / beta for x > 0, else 0; beta = 20^4 */
#define EDET_ACT_LAST EDET_ACT_SRELU

/* resample modes of one BiFPN fusion input */
#define EDET_RS_IDENTITY 0
#define EDET_RS_UP2 1   /* nearest-neighbour upsample, src = min(floor(dst*in/out), in-1) */
#define EDET_RS_POOL 2  /* max-pool 3x3 stride 2, TF 'SAME', padding excluded from the max */

/* number of per-workgroup statistic partial rows a kernel may write */
#define EDET_MAX_PARTS 1024

/*
 * "Activated view" of a stored tensor: value(n,h,w,c) =
 *     act(data * scale[c] + shift[c]) * gate[n,c]
 * scale/shift NULL -> identity affine; gate NULL -> 1.  This is how BatchNorm
 * (utils.py:166-266), swish (utils.py:36-39) and the SE gate
 * (efficientnet_model.py:183-195) are applied on load by the consuming kernel
 * instead of as separate HBM passes.
 */
typedef struct edet_tview {
  const void* data;
  const float* scale;
  const float* shift;
  const float* gate;
  int act;
  int n, h, w, c, ld;
} edet_tview_t;

/*
 * Gradient view: dy(n,h,w,c) = a[c]*dz + b[c]*y + cc[c]  (a == NULL -> dy = dz).
 * This is the BatchNorm backward applied on load: dz is the gradient w.r.t. the
 * BN output, y the saved output of THE convolution whose gradients are being taken
 * (see EDET_EPI_Y_IS_CONV_OF_INPUT), and (a,b,cc) come from edet_bn_bwd_finalize.
 */
typedef struct edet_gview {
  const void* dz;
  const void* y;
  const float* a;
  const float* b;
  const float* cc;
  int n, h, w, c, ld;
} edet_gview_t;

/*
 * What a data-gradient kernel does with d(view) for its input view `in`:
 *   plain        : g = d
 *   act          : g = d * act'(z),  z = data*scale+shift
 *   gate         : store D = d (the gated gradient) and accumulate
 *                  dgate[n,c] += sum_hw D * act(z); edet_se_gate_bwd finishes.
 *                  (beta must be 0 with a gate: the fp32 / generic kernel takes
 *                  the gate sums from the gradient it has just stored, in a
 *                  fixed order -- no atomics; an SE output has one consumer.)
 *   beta != 0    : g += previous contents of gout.
 *   stat_partials: per-workgroup partial sums of (g, g*xhat), xhat =
 *                  (data-mean)*rstd, row p at stat_partials[p*2*c .. ]; the
 *                  kernel writes exactly `*nparts_out` rows.
 */
typedef struct edet_bwd_epi {
  void* gout;
  int beta;
  const float* mean;
  const float* rstd;
  float* stat_partials;
  float* dgate;
  int flags;      /* EDET_EPI_* bits */
} edet_bwd_epi_t;
/* The convolution whose output gradient `dy` is has NO bias, so dy->y (its saved output) equals view(in) * W as
 * edet_pw_fwd stored it.  With this bit set the pointwise backward entry points may apply the b*y term of the
 * BatchNorm backward through the convolution INPUT (dx = dz (W diag a)^T + x~ (W diag(b) W^T) + c W^T, dW = (x~^T dz)
 * diag a + (x~^T x~) W diag b + (sum x~) c^T) and never read y: for an MBConv expansion that is 43 % less HBM traffic.
 * Without it they read y.  Valid only while the stored y is exactly what edet_pw_fwd wrote from THIS input view and THIS
 * kernel (no in-place modification of x, W or y in between); the recomputed product is not rounded to the storage type,
 * so the gradients differ from the read-y form by the rounding of y (2^-9 relative per element, zero mean) --
 * tests/test_gpu_kernels.py::test_pw_bwd runs both forms (EDET_PW_NOY=0 / 1) against the oracle under this contract.  */
#define EDET_EPI_Y_IS_CONV_OF_INPUT 1

const char* edet_last_error(void);
int edet_version(void);

/* ---- debug launch log ----------------------------------------------------------
 * Test infrastructure on the product side of the boundary: while the log is on, the library counts every
 * kernel launch by kernel symbol.  edet_debug_launch_log(1) clears the log and starts it, (0) stops it;
 * edet_debug_launch_names writes "count<TAB>demangled kernel name<NEWLINE>" lines (NUL terminated, truncated
 * to `capacity`; *needed = bytes for the whole text).  tests/test_gpu_bench_shapes.py uses it to assert that
 * every kernel symbol of the full-size benchmark step is also launched by a test that checks results
 * against the oracle.  The reference has no counterpart (TensorFlow picks its kernels internally).  */
int edet_debug_launch_log(int enable);
int edet_debug_launch_names(char* buf, size_t capacity, size_t* needed);

/* ---- parameter preparation ------------------------------------------------
 * fp32 master -> compute copy (dtype), optionally transposed [rows][cols] ->
 * [cols][ld_out].  Replaces the Keras mixed-precision variable cast
 * (utils.py:552-566).  */
int edet_cast(const float* src, void* dst, int64_t count, int dtype, void* stream);
int edet_cast_matrix(const float* src, void* dst, int rows, int cols, int ld_out,
                     int transpose, int dtype, void* stream);

/* all compute copies of a step in ONE launch: items_dev is a device array of `count` descriptors (the
 * arguments of edet_cast_matrix), max_blocks_per_item bounds the 256-thread blocks that stride over one item.  */
typedef struct edet_cast_item {
  const float* src;
  void* dst;
  int rows, cols, ld_out, transpose;
} edet_cast_item_t;
int edet_cast_batch(const edet_cast_item_t* items_dev, int count, int max_blocks_per_item, int dtype,
                    void* stream);

/* ---- stem: Conv2D 3x3 stride 2 'SAME', Cin = 3, no bias --------------------
 * efficientnet_model.py:511-519.  images [n,h,w,3] (ld 3), weight fp32 HWIO
 * [3,3,3,cout].  Writes raw conv output and BN statistic partials.  */
int edet_stem_fwd(const void* images, int n, int h, int w, const float* weight,
                  void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                  int dtype, void* stream);
/* dweight [3,3,3,cout] fp32 is accumulated into.  workspace: caller-owned device scratch for the per-workgroup
 * partial sums (27 * cout floats each, at most EDET_MAX_PARTS of them), added in a fixed order -- the same
 * gradient on every run; may be NULL (or smaller than EDET_MAX_PARTS rows), then ONE workgroup computes the whole sum
 * and adds it into dweight itself (slow, still the same bits on every run: the library has no floating-point atomics
 * on this path in either storage type).  */
int edet_stem_bwd_weight(const void* images, int n, int h, int w,
                         const edet_gview_t* dy, float* dweight, void* workspace, size_t workspace_bytes,
                         int dtype, void* stream);

/* ---- pointwise (1x1) convolution = GEMM on the matrix cores ----------------
 * Conv2D 1x1 call sites: efficientnet_model.py:304-312,345-353;
 * efficientdet_keras.py:286-290 and the pointwise half of SeparableConv2D
 * (:195-207,459-464,546-556).  wt is the compute copy [cout][ldw] (K contiguous),
 * bias fp32 [cout] or NULL.  */
int edet_pw_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias,
                void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                int dtype, void* stream);
/* the same 1x1 convolution of a bf16 view with the output stored as fp32 [rows][ldo] (ldo in floats, a multiple of 8,
 * >= cout), no statistics: the class / box predict layers of the inference forward (efficientdet_keras.py:459-464,
 * 546-556) -- rounding the logits to bf16 alone costs 3e-3 of their range, the convolution itself 1e-4 */
int edet_pw_fwd_f32out(const edet_tview_t* in, const void* wt, int ldw, const float* bias, float* out,
                       int cout, int ldo, void* stream);
/* d(in) from dy: w is the compute copy [cin][ldw] (Cout contiguous).  */
int edet_pw_bwd_data(const edet_gview_t* dy, const void* w, int ldw,
                     const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                     int dtype, void* stream);
/* dweight[cin][cout] (fp32, HWIO of a 1x1 kernel) += in^T dy.
 * workspace: caller-owned device scratch (fp32 partial sums [splits][cin][cout], deterministic
 * two-pass reduction); may be NULL -- the fp32 / generic kernel then runs a single row split that adds into dweight
 * itself (slow, no atomics, the same bits on every run).  64 MiB covers every EfficientDet-D0..D7x layer at full speed.  */
int edet_pw_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, float* dweight,
                       void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* both gradients of one layer in one call (TF's Conv2DBackpropInput + Conv2DBackpropFilter under the reference's
 * GradientTape, train_lib.py:623-669): the same result as edet_pw_bwd_weight followed by edet_pw_bwd_data with the
 * same arguments (workspace required).  In bf16, "expand"-shaped layers whose weight matrix fits a wave's registers
 * (cout >= 2 cin, cin <= 32, cout <= 144: the first MBConv expansions, where the 6x expanded gradient pair
 * dominates the traffic) run ONE fused kernel that reads dy, the saved convolution output behind it and the saved
 * input once; other shapes run the two kernels.  */
int edet_pw_bwd(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                size_t workspace_bytes, int dtype, void* stream);

/* ---- dense k x k convolution, k in {1,3,5}, stride in {1,2}, TF 'SAME', no bias ----
 * Fused-MBConv call sites: efficientnetv2/effnetv2_model.py:338-346 (k x k expand conv) and
 * :362-371 (the single k x k conv of an expand_ratio == 1 block).  wt is the compute copy
 * [cout][ldw] with the reduction index (ky*k + kx)*cin + c contiguous (edet_cast_matrix of the
 * HWIO kernel viewed as [k*k*cin][cout], transposed; ldw >= k*k*cin).  Writes the raw conv output
 * and BatchNorm statistic partials like edet_pw_fwd.  */
int edet_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int stride,
                  void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                  int dtype, void* stream);

/* gradients of the dense convolution (TF Conv2DBackpropInput / Conv2DBackpropFilter under the reference's
 * GradientTape).  w_t: compute copy [cin][ldw] with the reduction index (ky*k + kx)*cout + co contiguous (the HWIO
 * kernel permuted to [cin][k][k][cout]); epi as for edet_pw_bwd_data (no gate).  dweight: fp32 HWIO
 * [k][k][cin][cout], accumulated into; workspace as for edet_pw_bwd_weight.  */
int edet_conv_bwd_data(const edet_gview_t* dy, const void* w_t, int ldw, int k, int stride,
                       const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                       int dtype, void* stream);
int edet_conv_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, int k, int stride,
                         float* dweight, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ---- head of an MBConv block in one kernel (bf16 storage) ----------------------------------
 * efficientnet_model.py:378-392: x = act(bn0(expand_conv(x))); x = act(bn1(depthwise_conv(x))) -- the layers of
 * :304-327.  The expanded tensor (6 x the block input) is produced by the matrix cores INSIDE the depthwise march
 * instead of being written by edet_pw_fwd and read back by edet_dw_fwd (automl_amd/csrc/mbconv_fused.hip).
 *   in           : the block input as an affine view (the producer's BatchNorm on load, or a stored tensor): no
 *                  activation, no gate, c <= 32 channels (c % 8 == 0).
 *   w_t          : the expansion kernel as edet_pw_fwd takes it -- compute copy [cexp][ldw], input channel contiguous;
 *                  cexp % 48 == 0 (every EfficientNet expansion width is).
 *   exp_scale/exp_shift : the expansion's BatchNorm as edet_bn_finalize / edet_bn_eval wrote it; act: its activation.
 *   expanded_out : NULL (inference: the expanded tensor is never stored) or the raw expanded tensor [n,h,w,lde], which
 *                  the training backward pass reads (edet_dw_bwd, edet_pw_bwd) exactly as if edet_pw_fwd had stored it.
 *   out / stat_partials / nparts_out : as edet_dw_fwd.
 * Training needs the batch statistics of the expansion before the depthwise convolution can run:
 * edet_mbconv_expand_stats makes the partial rows (same matrix-core products, same bf16 rounding, nothing stored) that
 * edet_bn_finalize turns into exp_scale / exp_shift.  edet_mbconv_fused_supported: 1 = these two entry points apply to
 * the layer, 0 = the caller runs edet_pw_fwd + edet_dw_fwd.  */
int edet_mbconv_fused_supported(const edet_tview_t* in, int cexp, int k, int stride, int dtype);
int edet_mbconv_expand_stats(const edet_tview_t* in, const void* w_t, int ldw, int cexp,
                             float* stat_partials, int* nparts_out, int dtype, void* stream);
int edet_mbconv_expand_dw_fwd(const edet_tview_t* in, const void* w_t, int ldw, int cexp,
                              const float* exp_scale, const float* exp_shift, int act,
                              void* expanded_out, int lde, const float* dw_weight, int k, int stride,
                              void* out, int ldo, float* stat_partials, int* nparts_out, int dtype,
                              void* stream);

/* ---- depthwise convolution k in {3,5}, stride in {1,2}, TF 'SAME' ----------
 * DepthwiseConv2D call sites: efficientnet_model.py:320-327 and the depthwise
 * half of SeparableConv2D.  weight fp32 [k,k,c] (HWIO with multiplier 1).  */
int edet_dw_fwd(const edet_tview_t* in, const float* weight, int k, int stride,
                void* out, int ldo, float* stat_partials, int* nparts_out,
                int dtype, void* stream);
int edet_dw_bwd_data(const edet_gview_t* dy, const float* weight, int k, int stride,
                     const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                     int dtype, void* stream);
/* dweight [k,k,c] fp32 is accumulated into; workspace as for edet_pw_bwd_weight (may be NULL). */
int edet_dw_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, int k, int stride,
                       float* dweight, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* both gradients of one layer in one call: the same result as edet_dw_bwd_weight followed by
 * edet_dw_bwd_data (same arguments); for stride 1 in bf16 a single fused kernel that reads dy, the
 * saved conv output behind it and the saved input once.  */
int edet_dw_bwd(const edet_gview_t* dy, const float* weight, int k, int stride,
                const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                float* dweight, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ---- deferred weight-gradient reductions -------------------------------------------------
 * Every weight-gradient entry point above that takes a workspace ends in "dweight += sum of the partial sums it left
 * there" (a small kernel per call: ~190 of them per EfficientDet-D0 step).  edet_reduce_defer(stream, 1): from now on those
 * sums on `stream` are recorded instead of launched; edet_reduce_flush(stream) adds everything recorded in ONE launch
 * (same destinations in recording order -- a fixed summation order, the same result on every run; legal inside a
 * stream capture).  Between a call and the flush the caller must not reuse the part of the workspace that holds its partial
 * sums: edet_reduce_deferred_end gives the highest address of the recorded partial regions (NULL: nothing recorded), so a
 * caller can hand the next call the workspace from there on.  edet_reduce_defer(stream, 0) flushes and returns to the
 * immediate mode.  Host-side state per stream; not thread safe.  */
int edet_reduce_defer(void* stream, int enable);
int edet_reduce_flush(void* stream);
int edet_reduce_deferred_end(void* stream, const void** hi_out);

/* ---- BatchNorm statistics --------------------------------------------------
 * utils.py:244-266 / util_keras.py:29-66 (eps 1e-3, momentum 0.99).
 * finalize: partial sums -> batch mean / biased variance -> scale, shift, mean,
 * rstd; moving statistics updated in place when momentum >= 0: moving_var with the Bessel-corrected batch
 * variance when bessel != 0 (Keras' fused BatchNormalization, the single-replica classes of utils.py:244-266),
 * with the biased one when bessel == 0 (SyncBatchNormalization / TpuBatchNormalization force fused=False,
 * utils.py:166-213).  */
int edet_bn_finalize(const float* partials, int nparts, int c, double count,
                     const float* gamma, const float* beta, float eps, float momentum, int bessel,
                     float* moving_mean, float* moving_var,
                     float* scale, float* shift, float* mean, float* rstd, void* stream);
/* inference: scale/shift from the moving statistics */
int edet_bn_eval(int c, const float* gamma, const float* beta, float eps,
                 const float* moving_mean, const float* moving_var,
                 float* scale, float* shift, void* stream);
/* partial sums of (dz, dz*xhat) over a stored gradient (multi-consumer case) */
int edet_bn_bwd_reduce(const void* dz, const void* y, int64_t rows, int c, int ld,
                       const float* mean, const float* rstd,
                       float* stat_partials, int* nparts_out, int dtype, void* stream);
/* partials -> dgamma, dbeta (accumulated) and the on-load coefficients a,b,cc.
 * dbias is ignored (may be NULL): the gradient of a bias that feeds a BatchNorm is
 * analytically zero because BN removes the per-channel mean.  */
int edet_bn_bwd_finalize(const float* partials, int nparts, int c, double count,
                         const float* gamma, const float* mean, const float* rstd,
                         float* dgamma, float* dbeta, float* dbias,
                         float* a, float* b, float* cc, void* stream);

/* ---- block output: out = view(y) (+ residual) ---------------------------------
 * efficientnet_model.py:393-410 (project BN, identity skip).  y->gate [n][c], when set, is the
 * stochastic-depth scale floor(p + u_n) / p of utils.drop_connect (utils.py:329-344), constant along c.  */
int edet_bn_res(const edet_tview_t* y, const void* residual, void* out, int ldo,
                int dtype, void* stream);
/* dst = (beta ? dst : 0) + src, elementwise on [rows][c] with ld */
int edet_add(void* dst, const void* src, int64_t rows, int c, int ld, int beta,
             int dtype, void* stream);

/* ---- squeeze-and-excitation --------------------------------------------------
 * efficientnet_model.py:153-195: mean over H,W -> 1x1 (+bias) -> act (the model's relu_fn, an EDET_ACT_* code)
 * -> 1x1 (+bias) -> sigmoid.
 * edet_se_pool: pooled_sum [n,c] = sum over H,W of view(in) (the gate of `in` is ignored).  No atomics: an image's rows
 * are summed in chunks whose size depends on (H*W, c) only, the chunk sums (written to `scratch`, caller-owned, at
 * least n * ceil(H*W / chunk) * c floats: 8 MB covers every EfficientDet / EfficientNetV2 layer; the chunks grow if it
 * is smaller) are added in chunk order -- the same image gives the same bits in every batch and on every run.
 * edet_se_squeeze_excite: the pooling and both 1x1 layers in two launches (the FC kernel adds the chunk sums itself);
 * also writes pooled_sum (edet_se_fc_bwd reads it).  inv_hw = 1 / (H*W).  */
int edet_se_pool(const edet_tview_t* in, float* pooled_sum, void* scratch, size_t scratch_bytes,
                 int dtype, void* stream);
int edet_se_fc(const float* pooled_sum, int n, int c, int se, float inv_hw,
               const float* w1, const float* b1, const float* w2, const float* b2,
               float* hidden_pre, float* gate, int act, void* stream);
int edet_se_squeeze_excite(const edet_tview_t* in, void* scratch, size_t scratch_bytes, int se, float inv_hw,
                           const float* w1, const float* b1, const float* w2, const float* b2,
                           float* pooled_sum, float* hidden_pre, float* gate, int act, int dtype, void* stream);
/* dgate [n,c] -> dpool [n,c] (already divided by H*W), parameter gradients.
 * scratch: caller-owned fp32 workspace of n * (c + (2 + ceil(c / 128)) * se) + 8 * (2 * c * se + c + se) elements
 * (the second term: the parameter gradients of 8 image slices, added in slice order -- no atomics).  */
int edet_se_fc_bwd(const float* pooled_sum, const float* hidden_pre, const float* gate,
                   const float* dgate, int n, int c, int se, float inv_hw,
                   const float* w1, const float* w2,
                   float* dw1, float* db1, float* dw2, float* db2,
                   float* dpool, float* scratch, int act, void* stream);
/* in place on g (holding the gated gradient D): dz = (D*gate + dpool)*act'(z);
 * writes BN backward partials for `in`'s BatchNorm.  */
int edet_se_gate_bwd(const edet_tview_t* in, void* g, const float* dpool,
                     const float* mean, const float* rstd,
                     float* stat_partials, int* nparts_out, int dtype, void* stream);

/* ---- BiFPN weighted fusion ---------------------------------------------------
 * efficientdet_keras.py:75-121 (fuse_features), :254-281 (max-pool / nearest
 * resample), :214-217 (swish before the separable conv).
 * out = act( sum_i wn_i * resample_i(view_i) ), wn = normalised weights, fp32 [3][wc]: wc = 1 for one weight per
 * input (fastattn / attn / sum), wc = c for the per-channel methods channel_fastattn / channel_attn
 * (efficientdet_keras.py:100-113: the same two formulas applied per channel; WSM variables of shape [c]).  */
int edet_fuse_weights(const float* w0, const float* w1, const float* w2, int nin,
                      int method /*0 fastattn, 1 sum, 2 attn (softmax)*/, float* wn, int wc, void* stream);
/* wraw (may be NULL): the raw fusion variables {w0, w1, w2} (device scalars) when wc == 1 -- the kernel normalises them
 * itself with edet_fuse_weights' arithmetic (method as there) and stores the result in wn for the backward calls, so
 * the edet_fuse_weights launch is not needed; NULL or wc > 1: wn is read as computed by edet_fuse_weights.  */
int edet_fuse_fwd(const edet_tview_t* in0, const edet_tview_t* in1, const edet_tview_t* in2,
                  const int* modes, int nin, float* wn, int wc, int act,
                  void* out, int oh, int ow, int ldo, const float* const* wraw, int method, int dtype, void* stream);
/* ds = dout * act'(s) (s recomputed) written to `ds`; dwn[i] += sum ds * x_i.
 * gin / gbeta (may be NULL): per input, the gradient buffer of an EDET_RS_IDENTITY input that this call writes itself
 * (gin[i] (+)= wn[i] * ds when gbeta[i]; exactly what edet_fuse_bwd_input gives) -- saves that launch and its read of
 * ds; write_ds = 0 when no other input needs the stored ds.
 * workspace (may be NULL): caller-owned scratch for the per-workgroup partial sums of the scalar fusion weights
 * (16 bytes per workgroup: 64 KiB is enough; per-channel weights: nin * c floats per workgroup), added in a fixed order
 * -- the same dwn on every run; NULL or too small: the kernel runs as ONE workgroup that adds its sums into dwn itself
 * (slow, no atomics).
 * pool_argmax (may be NULL): caller-owned bytes [npool][n][oh][ow][c], one plane per EDET_RS_POOL
 * input in input order; receives the winning tap (ky*3+kx, first maximum of the row-major scan) of
 * every pooled element so that edet_fuse_bwd_input does not have to recompute the 3x3 windows.
 * wraw / dwraw (may be NULL): the raw scalar fusion variables and their gradients {dw0, dw1, dw2}; with both (and
 * wc == 1, a workspace, method 0 or 2) the ordered finish of dwn also adds the normalisation's backward into dwraw --
 * edet_fuse_weights_bwd's arithmetic without its launch (an error when the partial-row path is not available).  */
int edet_fuse_bwd_pre(const edet_tview_t* in0, const edet_tview_t* in1, const edet_tview_t* in2,
                      const int* modes, int nin, const float* wn, int wc, int act,
                      const void* dout, int oh, int ow, int ldo,
                      void* ds, float* dwn, void* pool_argmax, void* const* gin, const int* gbeta, int write_ds,
                      void* workspace, size_t workspace_bytes, const float* const* wraw, int method,
                      float* const* dwraw, int dtype, void* stream);
/* gradient of one fusion input: gout (+)= wn[i] * resample_i^T(ds).  pool_argmax: this input's
 * plane written by edet_fuse_bwd_pre (EDET_RS_POOL only; NULL -> the windows are recomputed).  */
int edet_fuse_bwd_input(const edet_tview_t* in, int mode, const float* wn, int wc, int idx,
                        const void* ds, int oh, int ow, int lds_, const void* pool_argmax,
                        void* gout, int beta, int dtype, void* stream);
/* raw weight gradients from dwn (fast-attention normalisation backward) */
int edet_fuse_weights_bwd(const float* w0, const float* w1, const float* w2, int nin,
                          int method, const float* dwn, float* dw0, float* dw1, float* dw2,
                          int wc, void* stream);

/* ---- detection loss forward + backward --------------------------------------
 * train_lib.py:357-437,493-604: focal loss (alpha, gamma) on class logits,
 * Huber(delta) on box codes; writes d(loss)/d(logits) and accumulates
 * sums[0] += cls_loss, sums[1] += box_loss (already normalised).
 * cls_targets int32 [n,h,w,a] (-1 background, -2 ignore).
 * norm_scale_dev (may be NULL): device scalar multiplied into inv_normalizer at run time, so that a
 * captured hipGraph of the step can be replayed with the next batch's normalizer
 * (sum(mean_num_positives) + 1, train_lib.py:517).
 * workspace: caller-owned device scratch for the per-workgroup partial rows (loss sum + bias gradient), added in a
 * fixed order -- the same loss and bias gradient on every run; (a few thousand rows of 1 + channels floats: 8 MiB
 * covers every EfficientDet head); NULL or too small: the kernel runs as ONE workgroup (slow, no atomics).  */
int edet_focal_loss(const void* logits, int ld, const int32_t* cls_targets,
                    int64_t positions, int num_anchors, int num_classes,
                    float alpha, float gamma, float inv_normalizer, const float* norm_scale_dev,
                    void* dlogits, float* dbias, float* sums, void* workspace, size_t workspace_bytes,
                    int dtype, void* stream);
/* the same with FocalLoss(label_smoothing) (train_lib.py:400-402, config.label_smoothing): the cross entropy is taken
 * against y*(1 - label_smoothing) + label_smoothing/2, alpha and the modulating factor keep the hard label */
int edet_focal_loss_smooth(const void* logits, int ld, const int32_t* cls_targets,
                           int64_t positions, int num_anchors, int num_classes,
                           float alpha, float gamma, float label_smoothing, float inv_normalizer,
                           const float* norm_scale_dev,
                           void* dlogits, float* dbias, float* sums, void* workspace, size_t workspace_bytes,
                           int dtype, void* stream);
int edet_box_loss(const void* box_out, int ld, const float* box_targets,
                  int64_t positions, int nch, float delta, float inv_normalizer,
                  float grad_scale, const float* norm_scale_dev, void* dbox, float* dbias, float* sums,
                  void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ---- optimizer -----------------------------------------------------------------
 * train_lib.py:486-491 (L2), :675-682 (per-tensor clip_by_norm then
 * clip_by_global_norm), Keras SGD momentum, TFA MovingAverage (:176-199).
 * All parameters live in one flat fp32 arena; seg_offsets[nseg+1] delimits the
 * tensors, seg_flags bit0 (EDET_SEG_L2) = L2-regularised (kernel/weight variables), bit1 (EDET_SEG_FROZEN) =
 * frozen by config.var_freeze_expr (tf2/train_lib.py:478-491: out of the L2 term, of the gradient list and of the
 * update -- its gradient is zeroed, it adds nothing to the norms, value / momentum / EMA shadow are never touched).  */
#define EDET_SEG_L2 1
#define EDET_SEG_FROZEN 2
/* seg_sqnorm: [2][nseg][EDET_OPT_SPLIT] floats -- EDET_OPT_SPLIT partial squared gradient norms per tensor and, behind
 * them, the tensors' partial L2 losses; written by edet_opt_l2_norms, summed in a fixed order by
 * edet_opt_clip_factors (large tensors are processed by up to EDET_OPT_SPLIT workgroups).  */
#define EDET_OPT_SPLIT 16
int edet_opt_l2_norms(float* grads, const float* params, const int64_t* seg_offsets,
                      const int32_t* seg_flags, int nseg, float weight_decay,
                      float* seg_sqnorm, void* stream);
/* seg_factor[s] = clip_by_norm factor * clip_by_global_norm factor; global_norm_out = norm after clipping;
 * l2_sum[0] += the L2 loss (either may be NULL) */
int edet_opt_clip_factors(const float* seg_sqnorm, int nseg, float clip_norm,
                          float* seg_factor, float* global_norm_out, float* l2_sum, void* stream);
/* grads[s] *= seg_factor[s]  (data-parallel path: clip locally, then all-reduce, train_lib.py:675-683) */
int edet_opt_scale(float* grads, const int64_t* seg_offsets, const float* seg_factor,
                   int nseg, void* stream);
/* hyper_dev = device float[2] {learning_rate, ema_decay}; seg_factor may be NULL (already scaled);
 * ema may be NULL; seg_flags may be NULL (no frozen variables).  v = momentum*v - lr*g; w += v;
 * ema -= (1-decay)*(ema - w); segments flagged EDET_SEG_FROZEN are skipped.  */
int edet_opt_sgd_ema(float* params, float* grads, float* velocity, float* ema,
                     const int64_t* seg_offsets, const float* seg_factor, const int32_t* seg_flags, int nseg,
                     const float* hyper_dev, float momentum, void* stream);
/* optimizer = 'adam' (train_lib.py:183-186: tf.keras.optimizers.Adam(learning_rate, beta_1=momentum), beta_2 0.999,
 * epsilon 1e-7): m, v = the two slot arenas; hyper_dev[0] = alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) of THIS step
 * (formed by the host: Engine.set_hyper), hyper_dev[1] = EMA decay; clip factors, frozen segments and the EMA as above */
int edet_opt_adam_ema(float* params, const float* grads, float* m, float* v, float* ema,
                      const int64_t* seg_offsets, const float* seg_factor, const int32_t* seg_flags, int nseg,
                      const float* hyper_dev, float beta1, float beta2, float epsilon, void* stream);

/* ---- step plumbing (round 6): the few operations of a step that are not layers, so that EVERY launch of a step goes
 * through this ABI and a step can be recorded and replayed without the Python interpreter (include/edet_net.h).
 * edet_zero: the start-of-step clears of the accumulation targets and the gradient arena (Engine._begin).
 * edet_axpy_clear: dst += src, then src = 0 when clear_src -- the side chain's gradient arena joined into the main one
 *   (Engine._join_side).
 * edet_loss_normalizer: inv_out[0] = 1 / (sum(mean_num_positives[0..n)) + 1), the per-step loss normalizer of
 *   tf2/train_lib.py:517-534 (positives_momentum = 0) kept on the device.  */
int edet_zero(void* dst, size_t bytes, void* stream);
/* dst [rows][c] = the first c elements of every row of src [rows][ld]: the level outputs without their padding columns, as
 * tf2/postprocess.py's reshape to [B, -1, num_classes] / [B, -1, 4] (:67-79) needs them (what `.contiguous()` did on the
 * Python host); rows of whole 4-byte words */
int edet_compact_rows(const void* src, int64_t rows, int c, int ld, void* dst, int elem_bytes, void* stream);
/* dst[i] = (float)src[i]: the fp32 copy of a stored tensor in front of a layer that runs in fp32 inside a bf16 network (the
 * box-predict island of the inference pass, Engine._to_f32); exact (bf16 -> fp32 widens) */
int edet_cast_to_f32(const void* src, float* dst, int64_t count, int src_dtype, void* stream);
int edet_axpy_clear(float* dst, float* src, int64_t n, int clear_src, void* stream);
int edet_loss_normalizer(const float* mean_num_positives, int n, float* inv_out, void* stream);

/* ---- detection post-processing -------------------------------
 * tf2/postprocess.py: merge_class_box_level_outputs :67-79, topk_class_boxes :82-117, pre_nms :120-157, nms :160-206,
 * clip_boxes :61-64, postprocess_global :375-406, per_class_nms :409-467; nms_np.py: hard_nms :84-120, soft_nms
 * :123-184, per_class_nms :214-265; tf2/anchors.py decode_box_outputs :30-58.
 *
 * edet_pre_nms: cls_levels[l] / box_levels[l] are the network outputs of level l, [batch][level_pixels[l]]
 * [anchors_per_pixel * num_classes] and [...][anchors_per_pixel * 4] (HOST arrays of DEVICE pointers); anchor_boxes
 * = the [N][4] fp32 anchors (ymin, xmin, ymax, xmax) in the order of anchors.Anchors.boxes.  Per anchor: the first
 * maximum class, sigmoid of its logit, the decoded box.  Outputs boxes [batch][N][4], scores [batch][N] fp32,
 * classes [batch][N] int32.  The first call for a given pyramid geometry uploads a (level, first anchor) table of a
 * few KB that stays cached on the device for the life of the process (one hipMalloc + one stream synchronisation);
 * every later call only launches.
 * edet_pre_nms_topk: the nms_configs.max_nms_inputs = k > 0 branch: the k largest (anchor, class) logits of every
 * image in descending order (ties: lower flat index), outputs [batch][k]...; k <= 8192.  */
int edet_pre_nms(const void* const* cls_levels, const void* const* box_levels, const int* level_pixels,
                 int nlevels, int batch, int anchors_per_pixel, int num_classes, const float* anchor_boxes,
                 int dtype, float* boxes, float* scores, int* classes, void* stream);
int edet_pre_nms_topk_workspace_bytes(int batch, int k, size_t* bytes);
int edet_pre_nms_topk(const void* const* cls_levels, const void* const* box_levels, const int* level_pixels,
                      int nlevels, int batch, int anchors_per_pixel, int num_classes, const float* anchor_boxes,
                      int dtype, int k, void* workspace, size_t workspace_bytes, float* boxes, float* scores,
                      int* classes, void* stream);

#define EDET_NMS_HARD 0
#define EDET_NMS_GAUSSIAN 1
#define EDET_NMS_LINEAR 2      /* nms_np.py only */
#define EDET_NMS_TF_V5 0       /* tf.raw_ops.NonMaxSuppressionV5: IoU on the corner extents, `score > score_thresh`
                                  filter up front, a decayed score <= score_thresh drops the candidate, sigma =
                                  soft_nms_sigma (= nms_configs.sigma / 2, postprocess.py:193-201) */
#define EDET_NMS_NUMPY 1       /* nms_np.py: pixel-inclusive extents (+1), no filter up front, a decayed score
                                  < score_thresh drops the candidate, weight exp(-iou^2 / sigma); hard = hard_nms */
typedef struct edet_nms_cfg {
  int method, convention;
  float iou_thresh, score_thresh, sigma;
  int max_output_size;
} edet_nms_cfg_t;
/* boxes [batch][n][4], scores [batch][n], classes [batch][n] (may be NULL when segments == 1).  segments == 1:
 * one suppression per image over all candidates (postprocess_global); segments == num_classes: one per (image,
 * class), merged per image to the max_output_size best by score (postprocess.per_class_nms, nms_np.per_class_nms).
 * out_index [batch][max_output_size]: index of the selected candidate or -1 (padding row), out_score: its decayed
 * score (0 for padding), out_valid [batch]: number of selected rows.  */
int edet_nms_workspace_bytes(int batch, int n, int segments, int max_output_size, size_t* bytes);
int edet_nms(const float* boxes, const float* scores, const int* classes, int batch, int n, int segments,
             const edet_nms_cfg_t* cfg, void* workspace, size_t workspace_bytes, int* out_index,
             float* out_score, int* out_valid, void* stream);
#define EDET_NMS_PAD_INDEX0 0  /* padding rows gather candidate 0 (tf.gather on the zero-padded indices), score 0 */
#define EDET_NMS_PAD_ZERO 1    /* padding rows are zeros (tf.pad in per_class_nms) */
#define EDET_NMS_PAD_DUMMY 2   /* padding rows are zeros with score -1e5 (nms_np._generate_dummy_detections) */
/* nms_boxes [batch][M][4] = boxes of the selected candidates, clipped to [0, clip_h] x [0, clip_w] when clip_h > 0
 * and multiplied by image_scales[b] when given; nms_classes = class + 1 (CLASS_OFFSET) as float.  */
int edet_nms_gather(const float* boxes, const int* classes, const int* out_index, const float* out_score,
                    int batch, int n, int max_output_size, int pad_mode, float clip_h, float clip_w,
                    const float* image_scales, float* nms_boxes, float* nms_scores, float* nms_classes,
                    void* stream);

/* ---- anchor labelling -------------------------------------------
 * tf2/anchors.py AnchorLabeler.label_anchors :215-250 for a batch of images.  anchor_boxes [N][4] as for
 * edet_pre_nms; level_anchors[l] = anchors of level l (H_l * W_l * A); gt_boxes [batch][max_gt][4] (ymin, xmin, ymax,
 * xmax), gt_labels [batch][max_gt] (1-based class ids), gt_count [batch] valid rows per image (device arrays).
 * Outputs per level (HOST arrays of DEVICE pointers): cls_targets[l] int32 [batch][H_l][W_l][A] (class - 1, -1 =
 * background), box_targets[l] fp32 [batch][H_l][W_l][4A]; num_positives fp32 [batch].  */
int edet_label_anchors_workspace_bytes(int batch, int num_anchors, size_t* bytes);
int edet_label_anchors(const float* anchor_boxes, const int* level_anchors, int nlevels, const float* gt_boxes,
                       const int* gt_labels, const int* gt_count, int batch, int max_gt, float match_threshold,
                       void* workspace, size_t workspace_bytes, int* const* cls_targets, float* const* box_targets,
                       float* num_positives, void* stream);

/* ---- inference image preprocessing ----------------------------------
 * efficientdet_keras.py:920-951 (mode 'infer'): raw_images [batch][height][width][3] uint8 (raw_is_float = 0) or
 * float32 (1) on the device, all of one size; out [batch][out_height][out_width][3] in `dtype`: normalised with
 * mean_rgb / stddev_rgb (HOST arrays of 3), aspect-preserving bilinear resize into the top-left corner, zero padding.
 * *image_scale_to_original (HOST) = 1 / scale, the factor that maps detections back to the raw image.  */
int edet_preprocess_infer(const void* raw_images, int raw_is_float, int batch, int height, int width,
                          int out_height, int out_width, const float* mean_rgb, const float* stddev_rgb, void* out,
                          float* image_scale_to_original, int dtype, void* stream);


/* ---- training image + box preprocessing --------------------------------
 * dataloader.py DetectionInputProcessor as InputReader.process_example drives it in training (:321-336): normalize_image
 * :58-64, random_horizontal_flip :150-153 (object_detection/preprocessor.py:113-199), set_training_random_scale_factors
 * :66-111, resize_and_crop_image :126-139, resize_and_crop_boxes :165-189 (clip_boxes :155-163, zero-area filter).
 * The random draws and the float32 scale arithmetic of set_training_random_scale_factors stay with the caller, who
 * hands over five integers per image (DEVICE array): flip decision, size of the resized image, crop offset.
 * raw_images as for edet_preprocess_infer.  boxes_in [batch][max_boxes][4] normalised (ymin, xmin, ymax, xmax),
 * classes_in [batch][max_boxes] float, counts_in [batch] valid rows; outputs: the kept boxes in pixels of the output
 * image and their classes IN ORDER, rows past counts_out[b] filled with -1 (dataloader.pad_to_fixed_size).  max_boxes
 * = 0 skips the box part (all box pointers may be NULL).  */
typedef struct edet_prep_image {
  int flip, scaled_h, scaled_w, offset_y, offset_x;
} edet_prep_image_t;
int edet_preprocess_train(const void* raw_images, int raw_is_float, int batch, int height, int width,
                          int out_height, int out_width, const float* mean_rgb, const float* stddev_rgb,
                          const edet_prep_image_t* per_image_dev, void* out, const float* boxes_in,
                          const float* classes_in, const int* counts_in, int max_boxes, float* boxes_out,
                          float* classes_out, int* counts_out, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* EDET_HIP_H_ */
