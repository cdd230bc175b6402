// Dense k x k convolution (stride 1 or 2, TF 'SAME', no bias): the Fused-MBConv convolutions of
// EfficientNetV2 (efficientnetv2/effnetv2_model.py:338-346 expand k x k, :362-371 single k x k conv).
//
// bf16: implicit GEMM on the matrix cores -- pw_big.hip's workgroup-tiled kernel with the row gather of the
// streamed operand replaced by the (ky, kx)-shifted input pixel (k_big_gemm<false, false, CONV=true>): rows =
// output pixels, reduction index = (ky*k + kx)*cin + c, BatchNorm / swish of the producer applied between the
// global load and the LDS store, zero 'SAME' padding inserted in the activated domain.  Arithmetic intensity
// is 9*cin*cout / (cin + cout) FLOP per bf16 element pair (200-900 FLOP/B for the V2-S stages): the only
// MFMA-bound kernel of this library; the 9-fold re-read of the input comes from L2 (rows of one row-tile
// group run back to back on one XCD).
// fp32 (validation mode) and shapes outside the envelope: a direct kernel, one thread per output pixel x 4
// output channels, weights read through L1 -- correct, not fast.
#include "common.h"

int cvh_try_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int s, void* out, int cout, int ldo,
                     float* stat_partials, int* nparts_out, hipStream_t st);
int pwb_try_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int s, const float* bias, void* out,
                     int cout, int ldo, float* stat_partials, int* nparts_out, hipStream_t st);

namespace {

struct ConvArgs {
  edet_tview_t in;
  const void* wt;     // [cout][ldw], reduction index (ky*k + kx)*cin + c contiguous
  int ldw, k, s, cout, ldo, oh, ow, pad_t, pad_l;
  void* out;
  float* stat_partials;
  int64_t M;
  int P;              // workgroups = stat partial rows
};

constexpr int DTHREADS = 256;

// thread -> (pixel p = tid / ncv, channel quad cv = tid % ncv); workgroup = DTHREADS / ncv pixels per step,
// marching over the pixels of its slice; per-channel sums of the rounded stored values are added through LDS, pixel
// lane after pixel lane, into one partial row per workgroup.
template <typename T>
__global__ __launch_bounds__(DTHREADS) void k_conv_direct(const ConvArgs a) {
  extern __shared__ float red[];      // [2][cout]
  const int ncv = (a.cout + 3) / 4;
  const int ppw = DTHREADS / ncv;     // pixels per workgroup step
  const int cv = threadIdx.x % ncv, pl = threadIdx.x / ncv;
  const int co = cv * 4;
  const T* X = reinterpret_cast<const T*>(a.in.data);
  const T* W = reinterpret_cast<const T*>(a.wt);
  T* O = reinterpret_cast<T*>(a.out);
  const int cin = a.in.c;
  const bool want_stats = a.stat_partials != nullptr;
  for (int i = threadIdx.x; i < 2 * a.cout; i += DTHREADS) red[i] = 0.f;
  __syncthreads();
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t per = (a.M + a.P - 1) / a.P;
  const int64_t mb = (int64_t)blockIdx.x * per, me = min(a.M, mb + per);
  if (pl < ppw) {
    for (int64_t m = mb + pl; m < me; m += ppw) {
      const int64_t img = m / ((int64_t)a.oh * a.ow);
      const int rem = (int)(m - img * a.oh * a.ow);
      const int oy = rem / a.ow, ox = rem - oy * a.ow;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ky = 0; ky < a.k; ++ky) {
        const int iy = oy * a.s - a.pad_t + ky;
        if (iy < 0 || iy >= a.in.h) continue;
        for (int kx = 0; kx < a.k; ++kx) {
          const int ix = ox * a.s - a.pad_l + kx;
          if (ix < 0 || ix >= a.in.w) continue;
          const T* xp = X + ((img * a.in.h + iy) * a.in.w + ix) * a.in.ld;
          const int kbase = (ky * a.k + kx) * cin;
          for (int c = 0; c < cin; ++c) {
            float x = to_f<T>(xp[c]);
            if (a.in.scale) x = fmaf(x, a.in.scale[c], a.in.shift[c]);
            x = act_apply_(a.in.act, x);
            if (a.in.gate) x *= a.in.gate[img * cin + c];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < a.cout) acc[e] = fmaf(x, to_f<T>(W[(size_t)(co + e) * a.ldw + kbase + c]), acc[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < a.cout) {
          const T o = from_f<T>(acc[e]);
          O[m * a.ldo + co + e] = o;
          const float v = to_f<T>(o);
          s1[e] += v;
          s2[e] = fmaf(v, v, s2[e]);
        }
      }
    }
  }
  if (want_stats) {
    // the ppw pixel lanes of a channel quad add their sums one after the other (a fixed order, no LDS atomics: the same
    // partial row on every run)
    for (int pv = 0; pv < ppw; ++pv) {
      if (pl == pv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < a.cout) {
            red[co + e] += s1[e];
            red[a.cout + co + e] += s2[e];
          }
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < 2 * a.cout; i += DTHREADS)
      a.stat_partials[(size_t)blockIdx.x * 2 * a.cout + i] = red[i];
  }
}

// ---- backward, direct forms (fp32 validation mode / shapes outside the tiled kernels' envelope) ----------------
struct ConvBwdArgs {
  edet_tview_t in;      // conv input view (must be plain or affine/act handled by the caller: see entry point)
  edet_gview_t gy;
  const void* w;        // dgrad: [cin][ldw] compute copy, index (ky*k+kx)*cout + co
  int ldw, k, s, oh, ow, pad_t, pad_l;
  void* gout;
  int beta;
  float* dweight;       // wgrad: fp32 HWIO
};

template <typename T>
__device__ __forceinline__ float conv_dy(const edet_gview_t& g, size_t off, int co) {
  float v = to_f<T>(reinterpret_cast<const T*>(g.dz)[off]);
  if (g.a) v = fmaf(g.a[co], v, fmaf(g.b[co], to_f<T>(reinterpret_cast<const T*>(g.y)[off]), g.cc[co]));
  return v;
}

// thread = (input pixel, input channel): d in = sum over taps whose (iy + pad - ky) / s is an integer dy row
template <typename T>
__global__ __launch_bounds__(DTHREADS) void k_conv_dgrad_direct(const ConvBwdArgs a) {
  const int cin = a.in.c, cout = a.gy.c;
  const int64_t total = (int64_t)a.in.n * a.in.h * a.in.w * cin;
  const T* W = reinterpret_cast<const T*>(a.w);
  T* GO = reinterpret_cast<T*>(a.gout);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin);
    const int64_t pix = i / cin;
    const int ix = (int)(pix % a.in.w), iy = (int)((pix / a.in.w) % a.in.h);
    const int64_t img = pix / ((int64_t)a.in.w * a.in.h);
    float acc = 0.f;
    for (int ky = 0; ky < a.k; ++ky) {
      const int ty = iy + a.pad_t - ky;
      if (ty < 0 || ty % a.s != 0 || ty / a.s >= a.oh) continue;
      for (int kx = 0; kx < a.k; ++kx) {
        const int tx = ix + a.pad_l - kx;
        if (tx < 0 || tx % a.s != 0 || tx / a.s >= a.ow) continue;
        const size_t base = ((size_t)(img * a.oh + ty / a.s) * a.ow + tx / a.s) * a.gy.ld;
        const T* wr = W + (size_t)c * a.ldw + (size_t)(ky * a.k + kx) * cout;
        for (int co = 0; co < cout; ++co) acc = fmaf(conv_dy<T>(a.gy, base + co, co), to_f<T>(wr[co]), acc);
      }
    }
    const size_t off = (size_t)pix * a.in.ld + c;
    if (a.beta) acc += to_f<T>(GO[off]);
    GO[off] = from_f<T>(acc);
  }
}

// thread = one weight (tap, c, co): sums over all output pixels (slow, exact order-independent fp32 sum per thread)
template <typename T>
__global__ __launch_bounds__(DTHREADS) void k_conv_wgrad_direct(const ConvBwdArgs a) {
  const int cin = a.in.c, cout = a.gy.c;
  const int total = a.k * a.k * cin * cout;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int co = i % cout, c = (i / cout) % cin, tap = i / (cout * cin);
  const int ky = tap / a.k, kx = tap - ky * a.k;
  const T* X = reinterpret_cast<const T*>(a.in.data);
  float acc = 0.f;
  for (int img = 0; img < a.in.n; ++img)
    for (int oy = 0; oy < a.oh; ++oy) {
      const int iy = oy * a.s - a.pad_t + ky;
      if (iy < 0 || iy >= a.in.h) continue;
      for (int ox = 0; ox < a.ow; ++ox) {
        const int ix = ox * a.s - a.pad_l + kx;
        if (ix < 0 || ix >= a.in.w) continue;
        float x = to_f<T>(X[((size_t)(img * a.in.h + iy) * a.in.w + ix) * a.in.ld + c]);
        if (a.in.scale) x = fmaf(x, a.in.scale[c], a.in.shift[c]);
        x = act_apply_(a.in.act, x);
        if (a.in.gate) x *= a.in.gate[(size_t)img * cin + c];
        acc = fmaf(x, conv_dy<T>(a.gy, ((size_t)(img * a.oh + oy) * a.ow + ox) * a.gy.ld + co, co), acc);
      }
    }
  a.dweight[i] += acc;
}

}  // namespace

int pwb_try_conv_dgrad(const edet_gview_t* dy, const void* w_t, int ldw, int k, int s, const edet_tview_t* in,
                       const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st);
int pwb_try_conv_wgrad(const edet_tview_t* in, const edet_gview_t* dy, int k, int s, float* dweight, void* workspace,
                       size_t workspace_bytes, hipStream_t st);

static int conv_bwd_common(ConvBwdArgs& a, const edet_tview_t* in, const edet_gview_t* dy, int k, int stride,
                           const char* who) {
  EDET_CHECK(k == 1 || k == 3 || k == 5, "%s: kernel size %d unsupported", who, k);
  EDET_CHECK(stride == 1 || stride == 2, "%s: stride %d unsupported", who, stride);
  memset(&a, 0, sizeof(a));
  a.in = *in; a.gy = *dy; a.k = k; a.s = stride;
  a.oh = same_out(in->h, stride); a.ow = same_out(in->w, stride);
  a.pad_t = same_pad_before(in->h, k, stride); a.pad_l = same_pad_before(in->w, k, stride);
  EDET_CHECK(a.oh == dy->h && a.ow == dy->w && dy->n == in->n, "%s: dy geometry does not match the input", who);
  return 0;
}

extern "C" int edet_conv_bwd_data(const edet_gview_t* dy, const void* w_t, int ldw, int k, int stride,
                                  const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                                  int dtype, void* stream) {
  EDET_CHECK(dy && dy->dz && w_t && in && in->data && epi && epi->gout, "edet_conv_bwd_data: null pointer");
  EDET_CHECK(dtype == EDET_BF16 || dtype == EDET_F32, "edet_conv_bwd_data: bad dtype %d", dtype);
  EDET_CHECK(ldw >= k * k * dy->c, "edet_conv_bwd_data: ldw too small");
  EDET_CHECK(!epi->dgate && !in->gate, "edet_conv_bwd_data: gated inputs are not supported");
  ConvBwdArgs a;
  if (int rc = conv_bwd_common(a, in, dy, k, stride, "edet_conv_bwd_data")) return rc;
  hipStream_t st = to_stream(stream);
  if (dtype == EDET_BF16) {
    const int rc = pwb_try_conv_dgrad(dy, w_t, ldw, k, stride, in, epi, nparts_out, st);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  // direct form: plain inputs only (every dense convolution of EfficientNetV2 reads a stored block output)
  EDET_CHECK(in->act == EDET_ACT_NONE && !in->scale && !epi->stat_partials,
             "edet_conv_bwd_data: the direct kernel needs a plain input view (no BN / activation chain)");
  a.w = w_t; a.ldw = ldw; a.gout = epi->gout; a.beta = epi->beta;
  const int64_t total = (int64_t)in->n * in->h * in->w * in->c;
  const int grid = (int)((total + DTHREADS - 1) / DTHREADS < 8192 ? (total + DTHREADS - 1) / DTHREADS : 8192);
  if (dtype == EDET_BF16) edet_launch(k_conv_dgrad_direct<bf16_t>, grid, dim3(DTHREADS), 0, st, a);
  else edet_launch(k_conv_dgrad_direct<float>, grid, dim3(DTHREADS), 0, st, a);
  if (nparts_out) *nparts_out = 0;
  EDET_LAUNCH_CHECK("edet_conv_bwd_data");
  return 0;
}

extern "C" int edet_conv_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, int k, int stride,
                                    float* dweight, void* workspace, size_t workspace_bytes, int dtype,
                                    void* stream) {
  EDET_CHECK(in && in->data && dy && dy->dz && dweight, "edet_conv_bwd_weight: null pointer");
  EDET_CHECK(dtype == EDET_BF16 || dtype == EDET_F32, "edet_conv_bwd_weight: bad dtype %d", dtype);
  ConvBwdArgs a;
  if (int rc = conv_bwd_common(a, in, dy, k, stride, "edet_conv_bwd_weight")) return rc;
  hipStream_t st = to_stream(stream);
  if (dtype == EDET_BF16) {
    const int rc = pwb_try_conv_wgrad(in, dy, k, stride, dweight, workspace, workspace_bytes, st);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  a.dweight = dweight;
  const int total = k * k * in->c * dy->c;
  if (dtype == EDET_BF16) edet_launch(k_conv_wgrad_direct<bf16_t>, dim3((total + DTHREADS - 1) / DTHREADS), dim3(DTHREADS), 0, st, a);
  else edet_launch(k_conv_wgrad_direct<float>, dim3((total + DTHREADS - 1) / DTHREADS), dim3(DTHREADS), 0, st, a);
  EDET_LAUNCH_CHECK("edet_conv_bwd_weight");
  return 0;
}

extern "C" int edet_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int stride,
                             void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                             int dtype, void* stream) {
  EDET_CHECK(in && in->data && wt && out, "edet_conv_fwd: null pointer");
  EDET_CHECK(k == 1 || k == 3 || k == 5, "edet_conv_fwd: kernel size %d unsupported", k);
  EDET_CHECK(stride == 1 || stride == 2, "edet_conv_fwd: stride %d unsupported", stride);
  EDET_CHECK(dtype == EDET_BF16 || dtype == EDET_F32, "edet_conv_fwd: bad dtype %d", dtype);
  EDET_CHECK(ldw >= k * k * in->c && ldo >= cout, "edet_conv_fwd: ldw/ldo too small");
  EDET_CHECK(cout <= 1024, "edet_conv_fwd: cout %d unsupported", cout);
  hipStream_t st = to_stream(stream);
  if (dtype == EDET_BF16) {
    // 3 x 3 stride 1 on 24 / 48 / 64 channels: from an LDS-resident halo tile (conv_halo.hip); else the implicit GEMM
    const int rh = cvh_try_conv_fwd(in, wt, ldw, k, stride, out, cout, ldo, stat_partials, nparts_out, st);
    if (rh != 0) return rh < 0 ? rh : 0;
    const int rc = pwb_try_conv_fwd(in, wt, ldw, k, stride, nullptr, out, cout, ldo, stat_partials, nparts_out, st);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.wt = wt; a.ldw = ldw; a.k = k; a.s = stride; a.cout = cout; a.ldo = ldo; a.out = out;
  a.stat_partials = stat_partials;
  a.oh = same_out(in->h, stride); a.ow = same_out(in->w, stride);
  a.pad_t = same_pad_before(in->h, k, stride); a.pad_l = same_pad_before(in->w, k, stride);
  a.M = (int64_t)in->n * a.oh * a.ow;
  const int ncv = (cout + 3) / 4, ppw = DTHREADS / ncv;
  int64_t P = (a.M + ppw - 1) / ppw;
  if (P > EDET_MAX_PARTS) P = EDET_MAX_PARTS;
  if (P < 1) P = 1;
  a.P = (int)P;
  if (nparts_out) *nparts_out = a.P;
  const size_t lds = 2 * (size_t)cout * sizeof(float);
  if (dtype == EDET_BF16) edet_launch(k_conv_direct<bf16_t>, dim3(a.P), dim3(DTHREADS), lds, st, a);
  else edet_launch(k_conv_direct<float>, dim3(a.P), dim3(DTHREADS), lds, st, a);
  EDET_LAUNCH_CHECK("edet_conv_fwd");
  return 0;
}
