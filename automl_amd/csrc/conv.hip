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
// marching over the pixels of its slice; per-channel sums of the rounded stored values are reduced through
// LDS atomics into one partial row per workgroup.
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
            if (a.in.act == EDET_ACT_SWISH) x = swishf_(x);
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
    if (pl < ppw) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e < a.cout) {
          atomicAdd(&red[co + e], s1[e]);
          atomicAdd(&red[a.cout + co + e], s2[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.cout; i += DTHREADS)
      a.stat_partials[(size_t)blockIdx.x * 2 * a.cout + i] = red[i];
  }
}

}  // namespace

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
  if (dtype == EDET_BF16) k_conv_direct<bf16_t><<<dim3(a.P), dim3(DTHREADS), lds, st>>>(a);
  else k_conv_direct<float><<<dim3(a.P), dim3(DTHREADS), lds, st>>>(a);
  EDET_LAUNCH_CHECK("edet_conv_fwd");
  return 0;
}
