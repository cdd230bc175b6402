// BiFPN weighted feature fusion with the resampling of its inputs folded in.
//
//   out = act( sum_i wn_i * R_i(view_i) ),  R in {identity, nearest-upsample, max-pool 3x3/s2 'SAME'}
//
// One elementwise pass: every input is read once (the pooled input 9 taps from L1/L2), BatchNorm
// of the producing 1x1 conv is applied on load, the fused + activated result is written once.
// Reference: efficientdet/tf2/efficientdet_keras.py:75-121 (fuse_features: fastattn / sum),
// :254-263 (MaxPooling2D pool=stride+1, 'SAME'), :272-281 (resize_nearest_neighbor),
// :214-217 (activation before the separable conv); efficientdet/efficientdet_arch.py:418-475.
#include "common.h"

namespace {

constexpr int THREADS = 256;

struct FuseArgs {
  edet_tview_t in[3];
  int mode[3];
  int pad_t[3], pad_l[3];
  int nin;
  const float* wn;  // [3][wc] normalised weights (device); wc = 1 (one weight per input) or c (per channel)
  int wc;
  int act;
  int n, oh, ow, c, ldo;
  // backward, r04: gradient buffers of EDET_RS_IDENTITY inputs that the kernel writes itself (NULL: left to
  // edet_fuse_bwd_input), accumulate flags, and whether ds has to be stored at all (another input still reads it)
  void* gin[3];
  int gbeta[3];
  int write_ds;
  // r04: the raw scalar fusion variables (wc == 1).  Forward: every workgroup normalises them itself (three scalars)
  // and workgroup 0 stores the result in wn_out for the backward kernels -- no k_fuse_weights launch.
  const float* wraw[3];
  float* wn_out;
  int method;
};

// normalised fusion weights of one channel from the raw variables (efficientdet_keras.py:84-113): method 1 = 'sum' (ones),
// 2 = 'attn' (softmax), otherwise 'fastattn' (relu / (sum + 1e-4)).  One code path for k_fuse_weights and k_fuse.
__device__ __forceinline__ void normalise_weights(const float (&w)[3], int nin, int method, float (&wn)[3]) {
  if (method == 1) {
    for (int i = 0; i < nin; ++i) wn[i] = 1.f;
    return;
  }
  if (method == 2) {      // 'attn': softmax over the inputs (efficientdet_keras.py:84-88)
    float m = w[0], e[3], s = 0.f;
    for (int i = 1; i < nin; ++i) m = fmaxf(m, w[i]);
    for (int i = 0; i < nin; ++i) { e[i] = expf(w[i] - m); s += e[i]; }
    for (int i = 0; i < nin; ++i) wn[i] = e[i] / s;
    return;
  }
  float r[3], s = 0.f;
  for (int i = 0; i < nin; ++i) { r[i] = fmaxf(w[i], 0.f); s += r[i]; }
  for (int i = 0; i < nin; ++i) wn[i] = r[i] / (s + 0.0001f);
}

// d(raw variables) += backward of normalise_weights given d(normalised weights) (method 1: nothing to do)
__device__ __forceinline__ void normalise_weights_bwd(const float (&w)[3], int nin, int method, const float (&dwn)[3],
                                                      float (&dw)[3]) {
  for (int i = 0; i < 3; ++i) dw[i] = 0.f;
  if (method == 1) return;
  if (method == 2) {      // softmax backward: dw_i = p_i * (dwn_i - sum_j dwn_j p_j)
    float m = w[0], p[3], s = 0.f, dot = 0.f;
    for (int i = 1; i < nin; ++i) m = fmaxf(m, w[i]);
    for (int i = 0; i < nin; ++i) { p[i] = expf(w[i] - m); s += p[i]; }
    for (int i = 0; i < nin; ++i) { p[i] /= s; dot += dwn[i] * p[i]; }
    for (int i = 0; i < nin; ++i) dw[i] = p[i] * (dwn[i] - dot);
    return;
  }
  float r[3], s = 0.0001f, dot = 0.f;
  for (int i = 0; i < nin; ++i) { r[i] = fmaxf(w[i], 0.f); s += r[i]; }
  for (int i = 0; i < nin; ++i) dot += dwn[i] * r[i];
  for (int i = 0; i < nin; ++i) {
    const float dr = dwn[i] / s - dot / (s * s);
    if (w[i] > 0.f) dw[i] = dr;
  }
}

// affine-only view value (fusion inputs never carry an activation or gate)
__device__ __forceinline__ void affine8(const edet_tview_t& v, const ViewCoef& k, float x[8]) {
  if (v.scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], k.scale[e], k.shift[e]);
  }
}

// value of resampled input i at output pixel (n, oy, ox), channels c0..c0+7
template <typename T, bool ARGMAX>
__device__ __forceinline__ void sample_input(const FuseArgs& a, int i, const ViewCoef& k, int n, int oy,
                                             int ox, int c0, float x[8], uint32_t (&am)[2]) {
  const edet_tview_t& v = a.in[i];
  const T* base = reinterpret_cast<const T*>(v.data);
  if (a.mode[i] == EDET_RS_IDENTITY) {
    load8<T>(base + ((size_t)(n * v.h + oy) * v.w + ox) * v.ld + c0, x);
    affine8(v, k, x);
  } else if (a.mode[i] == EDET_RS_UP2) {
    // nearest source pixel floor(o * in / out); exactly o >> 1 for the 2x pyramids
    int sy = a.oh == 2 * v.h ? oy >> 1 : (int)((uint32_t)(oy * v.h) / (uint32_t)a.oh);
    int sx = a.ow == 2 * v.w ? ox >> 1 : (int)((uint32_t)(ox * v.w) / (uint32_t)a.ow);
    sy = sy < v.h - 1 ? sy : v.h - 1;
    sx = sx < v.w - 1 ? sx : v.w - 1;
    load8<T>(base + ((size_t)(n * v.h + sy) * v.w + sx) * v.ld + c0, x);
    affine8(v, k, x);
  } else {  // max-pool 3x3 stride 2, padding excluded
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = -INFINITY;
    int idx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) idx[e] = 0;
    for (int ky = 0; ky < 3; ++ky) {
      const int sy = oy * 2 - a.pad_t[i] + ky;
      if (sy < 0 || sy >= v.h) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int sx = ox * 2 - a.pad_l[i] + kx;
        if (sx < 0 || sx >= v.w) continue;
        float t[8];
        load8<T>(base + ((size_t)(n * v.h + sy) * v.w + sx) * v.ld + c0, t);
        affine8(v, k, t);
        if (ARGMAX) {
          // first maximum of the row-major scan wins ties (the oracle's argmax convention)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool gt = t[e] > x[e];
            idx[e] = gt ? ky * 3 + kx : idx[e];
            x[e] = gt ? t[e] : x[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], t[e]);
        }
      }
    }
    if (ARGMAX) {
      am[0] = (uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24);
      am[1] = (uint32_t)idx[4] | ((uint32_t)idx[5] << 8) | ((uint32_t)idx[6] << 16) | ((uint32_t)idx[7] << 24);
    }
  }
}

// PCDW (backward, one fusion weight per channel, dwn wanted): thread = (8-channel chunk, pixel slice) for the whole
// kernel, so its dwn sums stay in registers; the slices are added through LDS in slice order and the workgroup's sums go
// to its partial row (edet_reduce_partials adds the rows in order) or, for a single workgroup, into dwn -- no atomics,
// the same gradient on every run.
// (one body, two kernel symbols: k_fuse<T, BWD> keeps its name and signature -- the committed kernel statistics and the
// coverage test of tests/test_gpu_bench_shapes.py name it -- and k_fuse_pc<T> is the PCDW backward)
template <typename T, bool BWD, bool PCDW>
__device__ __forceinline__ void fuse_body(const FuseArgs& a, T* __restrict__ out, const T* __restrict__ dout,
                                          T* __restrict__ ds, float* dwn, unsigned char* __restrict__ pool_argmax,
                                          float* dwn_parts) {
  const int nvec = a.c / 8;
  const int64_t total = (int64_t)a.n * a.oh * a.ow * nvec;
  const bool per_ch = a.wc > 1;                      // channel_attn / channel_fastattn: one weight per channel
  extern __shared__ float redc[];                    // PCDW: [slices][nin][c] dwn sums of this workgroup's pixel slices
  float wn[3] = {0.f, 0.f, 0.f};
  if (!BWD && !per_ch && a.wn_out) {
    float w[3] = {0.f, 0.f, 0.f};
    if (a.method != 1)
      for (int i = 0; i < a.nin; ++i) w[i] = a.wraw[i][0];
    normalise_weights(w, a.nin, a.method, wn);
    if (blockIdx.x == 0 && threadIdx.x < a.nin) a.wn_out[threadIdx.x] = wn[threadIdx.x];
  } else if (!per_ch) {
    for (int i = 0; i < a.nin; ++i) wn[i] = a.wn[i];
  }
  float dw_acc[3] = {0.f, 0.f, 0.f};
  float dwc[PCDW ? 3 : 1][8];
  // index arithmetic in 32 bits (round 6: the three int64 divisions per 16-byte chunk were most of the kernel's
  // instructions; the entry points refuse tensors of 2^31 chunks or more)
  uint32_t q0 = blockIdx.x * blockDim.x + threadIdx.x, qstride = gridDim.x * blockDim.x;
  const int slices = PCDW ? THREADS / nvec : 1;      // (host: c <= 8 * THREADS)
  if (PCDW) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) dwc[PCDW ? i : 0][e] = 0.f;
    const int chunk = threadIdx.x % nvec, slice = threadIdx.x / nvec;
    q0 = slice < slices ? (uint32_t)((blockIdx.x * slices + slice) * nvec + chunk) : (uint32_t)total;
    qstride = (uint32_t)(gridDim.x * slices * nvec);
  }
  const uint32_t total32 = (uint32_t)total, unvec = (uint32_t)nvec, uow = (uint32_t)a.ow, uoh = (uint32_t)a.oh;
  for (uint32_t q = q0; q < total32; q += qstride) {
    uint32_t pix = q / unvec;
    const int c0 = (int)(q - pix * unvec) * 8;
    const uint32_t prow = pix / uow;
    const int ox = (int)(pix - prow * uow);
    const int n = (int)(prow / uoh);
    const int oy = (int)(prow - (uint32_t)n * uoh);
    float s[8], xi[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const size_t opix = (size_t)(n * a.oh + oy) * a.ow + ox;
    int plane = 0;
    for (int i = 0; i < a.nin; ++i) {
      ViewCoef k;
      view_load_coef(a.in[i], c0, k);
      uint32_t am[2];
      if (BWD && pool_argmax && a.mode[i] == EDET_RS_POOL) {
        sample_input<T, true>(a, i, k, n, oy, ox, c0, xi[i], am);
        // plane p = [n][oh][ow][c] bytes: winning tap (ky*3+kx) of every pooled element
        *reinterpret_cast<uint2*>(pool_argmax + ((size_t)plane * a.n * a.oh * a.ow + opix) * a.c + c0) =
            make_uint2(am[0], am[1]);
        ++plane;
      } else {
        sample_input<T, false>(a, i, k, n, oy, ox, c0, xi[i], am);
      }
      if (per_ch) {
        float wv[8];
        loadf8(a.wn + (size_t)i * a.wc + c0, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = fmaf(wv[e], xi[i][e], s[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = fmaf(wn[i], xi[i][e], s[e]);
      }
    }
    const size_t off = opix * a.ldo + c0;
    if (!BWD) {
      if (a.act == EDET_ACT_SWISH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = swishf_(s[e]);
      } else if (a.act > EDET_ACT_SWISH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = act_other_(a.act, s[e]);
      }
      store8<T>(out + off, s);
    } else {
      float d[8];
      load8<T>(dout + off, d);
      if (a.act == EDET_ACT_SWISH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= swish_gradf_(s[e]);
      } else if (a.act > EDET_ACT_SWISH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= act_other_grad_(a.act, s[e]);
      }
      if (a.write_ds) store8<T>(ds + off, d);
      // identity inputs: g_i (+)= wn_i * ds at the same pixel, from the value ds is stored as (rounded to the storage
      // type first: the same bits edet_fuse_bwd_input would produce from the stored tensor)
      for (int i = 0; i < a.nin; ++i) {
        if (a.gin[i]) {
          float g[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = to_f<T>(from_f<T>(d[e]));
          if (per_ch) {
            float wv[8];
            loadf8(a.wn + (size_t)i * a.wc + c0, wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] *= wv[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] *= wn[i];
          }
          T* gp = reinterpret_cast<T*>(a.gin[i]) + opix * a.in[i].ld + c0;
          if (a.gbeta[i]) {
            float old[8];
            load8<T>(gp, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += old[e];
          }
          store8<T>(gp, g);
        }
      }
      for (int i = 0; i < a.nin; ++i) {
        if (per_ch) {
          if (PCDW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dwc[PCDW ? i : 0][e] = fmaf(d[e], xi[i][e], dwc[PCDW ? i : 0][e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dw_acc[i] = fmaf(d[e], xi[i][e], dw_acc[i]);
        }
      }
    }
  }
  if (PCDW) {
    const int chunk = threadIdx.x % nvec, slice = threadIdx.x / nvec;
    if (slice < slices) {
      for (int i = 0; i < a.nin; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) redc[((size_t)slice * a.nin + i) * a.c + chunk * 8 + e] = dwc[PCDW ? i : 0][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.nin * a.c; i += THREADS) {   // dwn [nin][wc], wc == c
      float t = 0.f;
      for (int sl = 0; sl < slices; ++sl) t += redc[(size_t)sl * a.nin * a.c + i];
      if (dwn_parts) dwn_parts[(size_t)blockIdx.x * a.nin * a.c + i] = t;
      else dwn[i] += t;
    }
  } else if (BWD && dwn && !per_ch) {
    // scalar fusion weights: wave shuffles, then the waves in order (r04: no LDS atomics); with a partial buffer the
    // workgroup's three sums go to its row there and k_fuse_dwn_finish adds the rows in order -- the same gradient on
    // every run -- without one the kernel is launched as ONE workgroup, which adds its sums into dwn
    __shared__ float red[THREADS / 64][4];
    for (int i = 0; i < a.nin; ++i) {
      float v = dw_acc[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      float t = 0.f;
      if (threadIdx.x < a.nin)
        for (int w = 0; w < THREADS / 64; ++w) t += red[w][threadIdx.x];
      if (dwn_parts) dwn_parts[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
      else if (threadIdx.x < a.nin) dwn[threadIdx.x] += t;       // (no partial buffer: launched as ONE workgroup)
    }
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(THREADS) void k_fuse(const FuseArgs a, T* __restrict__ out,
                                                 const T* __restrict__ dout, T* __restrict__ ds,
                                                 float* dwn, unsigned char* __restrict__ pool_argmax,
                                                 float* dwn_parts = nullptr) {
  fuse_body<T, BWD, false>(a, out, dout, ds, dwn, pool_argmax, dwn_parts);
}
template <typename T>
__global__ __launch_bounds__(THREADS) void k_fuse_pc(const FuseArgs a, T* __restrict__ out,
                                                    const T* __restrict__ dout, T* __restrict__ ds,
                                                    float* dwn, unsigned char* __restrict__ pool_argmax,
                                                    float* dwn_parts) {
  fuse_body<T, true, true>(a, out, dout, ds, dwn, pool_argmax, dwn_parts);
}

// dwn[i] += the workgroup rows of k_fuse<.., true>, in row order (i < 3); with the raw variables (r04) also their
// gradient dw_i += backward of the normalisation -- k_fuse_weights_bwd's arithmetic without its launch
struct FuseWBwd {
  const float* w[3];
  float* dw[3];
  int nin, method;
};
__global__ __launch_bounds__(256) void k_fuse_dwn_finish(const float* __restrict__ parts, int G, float* dwn, FuseWBwd wb) {
  __shared__ float red[256][3];
  __shared__ float tot[3];
  float t[3] = {0.f, 0.f, 0.f};
  // thread q adds the rows q*per .. (q+1)*per - 1 in order, thread 0 then adds the 256 chunk sums in order
  const int per = (G + 255) / 256;
  for (int g = threadIdx.x * per; g < min(G, (threadIdx.x + 1) * per); ++g)
    for (int i = 0; i < 3; ++i) t[i] += parts[(size_t)g * 4 + i];
  for (int i = 0; i < 3; ++i) red[threadIdx.x][i] = t[i];
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int q = 0; q < 256; ++q) s += red[q][threadIdx.x];
    dwn[threadIdx.x] += s;
    tot[threadIdx.x] = dwn[threadIdx.x];
  }
  if (wb.nin == 0 || wb.method == 1) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    float w[3] = {0.f, 0.f, 0.f}, d[3] = {tot[0], tot[1], tot[2]}, dw[3];
    for (int i = 0; i < wb.nin; ++i) w[i] = wb.w[i][0];
    normalise_weights_bwd(w, wb.nin, wb.method, d, dw);
    for (int i = 0; i < wb.nin; ++i)
      if (wb.method == 2 || w[i] > 0.f) wb.dw[i][0] += dw[i];
  }
}

// gradient w.r.t. the stored tensor of one fusion input (affine-only view => d(data) = scale * d(view);
// the scale factor is applied later by the BN backward coefficients, so here g = d(view))
struct FuseInArgs {
  edet_tview_t in;
  int mode, pad_t, pad_l;
  const float* wn;
  int wc;
  int idx;
  int n, oh, ow, ldds;
  int beta;
};

template <typename T>
__global__ __launch_bounds__(THREADS) void k_fuse_bwd_input(const FuseInArgs a, const T* __restrict__ ds,
                                                           T* __restrict__ gout,
                                                           const unsigned char* __restrict__ argmax) {
  const edet_tview_t& v = a.in;
  const int nvec = v.c / 8;
  const int64_t total = (int64_t)v.n * v.h * v.w * nvec;
  const float wn = a.wc > 1 ? 0.f : a.wn[a.idx];
  const T* base = reinterpret_cast<const T*>(v.data);
  const uint32_t total32 = (uint32_t)total, unvec = (uint32_t)nvec, uw = (uint32_t)v.w, uh = (uint32_t)v.h;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total32; q += gridDim.x * blockDim.x) {
    const uint32_t pix = q / unvec;
    const int c0 = (int)(q - pix * unvec) * 8;
    const uint32_t prow = pix / uw;
    const int sx = (int)(pix - prow * uw);
    const int n = (int)(prow / uh);
    const int sy = (int)(prow - (uint32_t)n * uh);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    if (a.mode == EDET_RS_IDENTITY) {
      load8<T>(ds + ((size_t)(n * a.oh + sy) * a.ow + sx) * a.ldds + c0, g);
    } else if (a.mode == EDET_RS_UP2) {
      // destination pixels whose nearest source is (sy, sx): oy with floor(oy*h/oh) == sy
      const bool x2 = a.oh == 2 * v.h && a.ow == 2 * v.w;      // the 2x pyramids: destination rows / columns 2s, 2s + 1
      const int oy_lo = x2 ? 2 * sy : (int)(((uint32_t)(sy * a.oh) + uh - 1) / uh);
      int oy_hi = x2 ? 2 * sy + 2 : (int)(((uint32_t)((sy + 1) * a.oh) + uh - 1) / uh);
      const int ox_lo = x2 ? 2 * sx : (int)(((uint32_t)(sx * a.ow) + uw - 1) / uw);
      int ox_hi = x2 ? 2 * sx + 2 : (int)(((uint32_t)((sx + 1) * a.ow) + uw - 1) / uw);
      if (sy == v.h - 1) oy_hi = a.oh;
      if (sx == v.w - 1) ox_hi = a.ow;
      for (int oy = oy_lo; oy < oy_hi; ++oy)
        for (int ox = ox_lo; ox < ox_hi; ++ox) {
          float t[8];
          load8<T>(ds + ((size_t)(n * a.oh + oy) * a.ow + ox) * a.ldds + c0, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] += t[e];
        }
    } else if (argmax) {
      // max-pool with the winners recorded by edet_fuse_bwd_pre: this pixel takes ds of every window
      // (at most 4) whose argmax tap is this pixel
      for (int oy = (sy + a.pad_t - 2 + 1) / 2; oy <= (sy + a.pad_t) / 2; ++oy) {
        if (oy < 0 || oy >= a.oh) continue;
        for (int ox = (sx + a.pad_l - 2 + 1) / 2; ox <= (sx + a.pad_l) / 2; ++ox) {
          if (ox < 0 || ox >= a.ow) continue;
          const uint32_t tap = (uint32_t)((sy - (oy * 2 - a.pad_t)) * 3 + (sx - (ox * 2 - a.pad_l)));
          const size_t opix = (size_t)(n * a.oh + oy) * a.ow + ox;
          const uint2 am = *reinterpret_cast<const uint2*>(argmax + opix * v.c + c0);
          float t[8];
          load8<T>(ds + opix * a.ldds + c0, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t w = e < 4 ? am.x : am.y;
            if (((w >> (8 * (e & 3))) & 0xffu) == tap) g[e] += t[e];
          }
        }
      }
    } else {
      // max-pool, winners recomputed: this pixel receives ds of every window in which it is the FIRST
      // maximum (row-major scan), matching the argmax convention of the oracle.
      ViewCoef k;
      view_load_coef(v, c0, k);
      float mine[8];
      load8<T>(base + ((size_t)(n * v.h + sy) * v.w + sx) * v.ld + c0, mine);
      affine8(v, k, mine);
      for (int oy = (sy + a.pad_t - 2 + 1) / 2; oy <= (sy + a.pad_t) / 2; ++oy) {
        if (oy < 0 || oy >= a.oh) continue;
        for (int ox = (sx + a.pad_l - 2 + 1) / 2; ox <= (sx + a.pad_l) / 2; ++ox) {
          if (ox < 0 || ox >= a.ow) continue;
          bool win[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) win[e] = true;
          for (int ky = 0; ky < 3; ++ky) {
            const int yy = oy * 2 - a.pad_t + ky;
            if (yy < 0 || yy >= v.h) continue;
            for (int kx = 0; kx < 3; ++kx) {
              const int xx = ox * 2 - a.pad_l + kx;
              if (xx < 0 || xx >= v.w) continue;
              if (yy == sy && xx == sx) continue;
              float t[8];
              load8<T>(base + ((size_t)(n * v.h + yy) * v.w + xx) * v.ld + c0, t);
              affine8(v, k, t);
              const bool before = (yy < sy) || (yy == sy && xx < sx);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                // an earlier element wins ties, a later one only if strictly greater
                if (before ? (t[e] >= mine[e]) : (t[e] > mine[e])) win[e] = false;
              }
            }
          }
          float t[8];
          load8<T>(ds + ((size_t)(n * a.oh + oy) * a.ow + ox) * a.ldds + c0, t);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (win[e]) g[e] += t[e];
        }
      }
    }
    const size_t off = ((size_t)(n * v.h + sy) * v.w + sx) * v.ld + c0;
    if (a.wc > 1) {
      float wv[8];
      loadf8(a.wn + (size_t)a.idx * a.wc + c0, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] *= wv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] *= wn;
    }
    if (a.beta) {
      float old[8];
      load8<T>(gout + off, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += old[e];
    }
    store8<T>(gout + off, g);
  }
}

// thread = channel ch < wc (wc = 1: the scalar weights of fastattn / attn; wc = c: channel_fastattn / channel_attn,
// efficientdet_keras.py:100-113): wn[i*wc + ch] from w_i[ch]
__global__ void k_fuse_weights(const float* w0, const float* w1, const float* w2, int nin, int method,
                               float* wn, int wc) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= wc) return;
  const float* wp[3] = {w0, w1, w2};
  float w[3] = {0.f, 0.f, 0.f}, r[3];
  if (method != 1)
    for (int i = 0; i < nin; ++i) w[i] = wp[i][ch];
  normalise_weights(w, nin, method, r);
  for (int i = 0; i < nin; ++i) wn[i * wc + ch] = r[i];
}

__global__ void k_fuse_weights_bwd(const float* w0, const float* w1, const float* w2, int nin, int method,
                                   const float* dwn, float* dw0, float* dw1, float* dw2, int wc) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= wc || method == 1) return;
  const float* wp[3] = {w0, w1, w2};
  float* dwp[3] = {dw0, dw1, dw2};
  float w[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f}, dw[3];
  for (int i = 0; i < nin; ++i) { w[i] = wp[i][ch]; d[i] = dwn[i * wc + ch]; }
  normalise_weights_bwd(w, nin, method, d, dw);
  for (int i = 0; i < nin; ++i)
    if (method == 2 || w[i] > 0.f) dwp[i][ch] += dw[i];
}

// grid of a grid-stride elementwise kernel: one thread per item up to ONE ROUND of what the chip holds of this kernel
// (occupancy query; 4096 when unknown) -- with more workgroups than resident slots the last round runs partly empty
inline int ew_grid(int64_t total, const void* fn = nullptr, size_t lds = 0) {
  int64_t g = (total + THREADS - 1) / THREADS;
  int cap = 4096;
  if (fn) {
    const int slots = edet_resident_wgs(fn, THREADS, lds);
    if (slots > 0) cap = slots;
  }
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int fill_args(FuseArgs& a, const edet_tview_t* in0, const edet_tview_t* in1, const edet_tview_t* in2,
              const int* modes, int nin, const float* wn, int wc, int act, int oh, int ow, int ldo) {
  EDET_CHECK(nin >= 1 && nin <= 3 && in0 && modes && wn, "edet_fuse: bad arguments");
  EDET_CHECK(wc == 1 || wc == in0->c, "edet_fuse: wc must be 1 or the channel count (got %d)", wc);
  const edet_tview_t* ins[3] = {in0, in1, in2};
  memset(&a, 0, sizeof(a));
  a.nin = nin; a.wn = wn; a.wc = wc; a.act = act; a.oh = oh; a.ow = ow; a.ldo = ldo;
  for (int i = 0; i < nin; ++i) {
    EDET_CHECK(ins[i] && ins[i]->data, "edet_fuse: null input %d", i);
    a.in[i] = *ins[i];
    a.mode[i] = modes[i];
    EDET_CHECK(ins[i]->act == EDET_ACT_NONE && !ins[i]->gate, "edet_fuse: inputs must be affine-only views");
    EDET_CHECK(ins[i]->c == in0->c && ins[i]->n == in0->n && ins[i]->c % 8 == 0 && ins[i]->ld % 8 == 0,
               "edet_fuse: channel/batch mismatch");
    if (modes[i] == EDET_RS_IDENTITY) {
      EDET_CHECK(ins[i]->h == oh && ins[i]->w == ow, "edet_fuse: identity input %d has a different size", i);
    } else if (modes[i] == EDET_RS_UP2) {
      EDET_CHECK(ins[i]->h <= oh && ins[i]->w <= ow, "edet_fuse: upsample input %d is larger than the output", i);
    } else if (modes[i] == EDET_RS_POOL) {
      EDET_CHECK(same_out(ins[i]->h, 2) == oh && same_out(ins[i]->w, 2) == ow,
                 "Incompatible Resampling : feat shape %dx%d target_shape: %dx%d", ins[i]->h, ins[i]->w, oh, ow);
      a.pad_t[i] = same_pad_before(ins[i]->h, 3, 2);
      a.pad_l[i] = same_pad_before(ins[i]->w, 3, 2);
    } else {
      EDET_CHECK(false, "edet_fuse: bad mode %d", modes[i]);
    }
  }
  a.n = in0->n; a.c = in0->c;
  EDET_CHECK(ldo % 8 == 0 && ldo >= a.c, "edet_fuse: bad ldo");
  return 0;
}

}  // namespace

extern "C" int edet_fuse_weights(const float* w0, const float* w1, const float* w2, int nin,
                                 int method, float* wn, int wc, void* stream) {
  EDET_CHECK(wn && (method == 1 || w0) && wc >= 1, "edet_fuse_weights: bad arguments");
  edet_launch(k_fuse_weights, dim3((wc + 63) / 64), dim3(64), 0, to_stream(stream), w0, w1, w2, nin, method, wn, wc);
  EDET_LAUNCH_CHECK("edet_fuse_weights");
  return 0;
}

extern "C" int edet_fuse_weights_bwd(const float* w0, const float* w1, const float* w2, int nin,
                                     int method, const float* dwn, float* dw0, float* dw1, float* dw2,
                                     int wc, void* stream) {
  if (method == 1) return 0;
  EDET_CHECK(w0 && dwn && dw0 && wc >= 1, "edet_fuse_weights_bwd: bad arguments");
  edet_launch(k_fuse_weights_bwd, dim3((wc + 63) / 64), dim3(64), 0, to_stream(stream), w0, w1, w2, nin, method, dwn, dw0, dw1, dw2, wc);
  EDET_LAUNCH_CHECK("edet_fuse_weights_bwd");
  return 0;
}

extern "C" int edet_fuse_fwd(const edet_tview_t* in0, const edet_tview_t* in1, const edet_tview_t* in2,
                             const int* modes, int nin, float* wn, int wc, int act,
                             void* out, int oh, int ow, int ldo, const float* const* wraw, int method, int dtype,
                             void* stream) {
  FuseArgs a;
  if (int rc = fill_args(a, in0, in1, in2, modes, nin, wn, wc, act, oh, ow, ldo)) return rc;
  EDET_CHECK(out, "edet_fuse_fwd: null output");
  if (wraw && wc == 1) {      // scalar fusion variables: normalised inside the kernel, stored to wn by workgroup 0
    EDET_CHECK(method >= 0 && method <= 2, "edet_fuse_fwd: bad method %d", method);
    for (int i = 0; i < nin; ++i) {
      EDET_CHECK(method == 1 || wraw[i], "edet_fuse_fwd: null fusion variable %d", i);
      a.wraw[i] = wraw[i];
    }
    a.wn_out = wn;
    a.method = method;
  }
  EDET_CHECK((int64_t)a.n * oh * ow * (a.c / 8) < (1ll << 31), "edet_fuse_fwd: tensor too large for the 32-bit index arithmetic");
  const int grid = ew_grid((int64_t)a.n * oh * ow * (a.c / 8), dtype == EDET_BF16 ? reinterpret_cast<const void*>(&k_fuse<bf16_t, false>) : nullptr);
  if (dtype == EDET_BF16) edet_launch(k_fuse<bf16_t, false>, grid, dim3(THREADS), 0, to_stream(stream), a, (bf16_t*)out, nullptr, nullptr, nullptr, nullptr, nullptr);
  else if (dtype == EDET_F32) edet_launch(k_fuse<float, false>, grid, dim3(THREADS), 0, to_stream(stream), a, (float*)out, nullptr, nullptr, nullptr, nullptr, nullptr);
  else EDET_CHECK(false, "edet_fuse_fwd: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_fuse_fwd");
  return 0;
}

extern "C" int edet_fuse_bwd_pre(const edet_tview_t* in0, const edet_tview_t* in1, const edet_tview_t* in2,
                                 const int* modes, int nin, const float* wn, int wc, int act,
                                 const void* dout, int oh, int ow, int ldo,
                                 void* ds, float* dwn, void* pool_argmax, void* const* gin, const int* gbeta,
                                 int write_ds, void* workspace, size_t workspace_bytes, const float* const* wraw,
                                 int method, float* const* dwraw, int dtype, void* stream) {
  FuseArgs a;
  if (int rc = fill_args(a, in0, in1, in2, modes, nin, wn, wc, act, oh, ow, ldo)) return rc;
  EDET_CHECK(dout && ds, "edet_fuse_bwd_pre: null pointer");
  a.write_ds = write_ds;
  for (int i = 0; i < nin; ++i) {
    a.gin[i] = gin ? gin[i] : nullptr;
    a.gbeta[i] = (gin && gbeta) ? gbeta[i] : 0;
    EDET_CHECK(!a.gin[i] || modes[i] == EDET_RS_IDENTITY, "edet_fuse_bwd_pre: gin[%d] given for a resampled input", i);
  }
  const bool pcdw = wc > 1 && dwn;
  EDET_CHECK(!pcdw || a.c <= 8 * THREADS, "edet_fuse_bwd_pre: per-channel fusion weights need c <= %d", 8 * THREADS);
  const size_t lds = pcdw ? (size_t)(THREADS / (a.c / 8)) * nin * a.c * sizeof(float) : 0;
  EDET_CHECK((int64_t)a.n * oh * ow * (a.c / 8) < (1ll << 31), "edet_fuse_bwd_pre: tensor too large for the 32-bit index arithmetic");
  int grid = ew_grid((int64_t)a.n * oh * ow * (a.c / 8), dtype == EDET_BF16 ? (pcdw ? reinterpret_cast<const void*>(&k_fuse_pc<bf16_t>) : reinterpret_cast<const void*>(&k_fuse<bf16_t, true>)) : nullptr, lds);
  // ordered partial rows through the workspace -- [grid][4] for scalar fusion weights, [grid][nin * c] for per-channel ones;
  // without a workspace that holds them ONE workgroup adds its sums into dwn (no atomics either way)
  const size_t row = pcdw ? (size_t)nin * a.c : 4;
  float* parts = (dwn && workspace && workspace_bytes >= (size_t)grid * row * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
  if (dwn && !parts) grid = 1;
  if (pcdw) {
    if (dtype == EDET_BF16) edet_launch(k_fuse_pc<bf16_t>, grid, dim3(THREADS), lds, to_stream(stream), a, nullptr, (const bf16_t*)dout, (bf16_t*)ds, dwn, (unsigned char*)pool_argmax, parts);
    else if (dtype == EDET_F32) edet_launch(k_fuse_pc<float>, grid, dim3(THREADS), lds, to_stream(stream), a, nullptr, (const float*)dout, (float*)ds, dwn, (unsigned char*)pool_argmax, parts);
    else EDET_CHECK(false, "edet_fuse_bwd_pre: bad dtype %d", dtype);
    EDET_LAUNCH_CHECK("edet_fuse_bwd_pre");
    // (edet_reduce_partials2: launched NOW, never recorded for a deferred batch -- edet_fuse_weights_bwd reads dwn next)
    if (parts && edet_reduce_partials2(parts, grid, (int64_t)nin * a.c, dwn, (int64_t)nin * a.c, nullptr, to_stream(stream)) != 0) return -2;
    return 0;
  }
  if (dtype == EDET_BF16) edet_launch(k_fuse<bf16_t, true>, grid, dim3(THREADS), lds, to_stream(stream), a, nullptr, (const bf16_t*)dout, (bf16_t*)ds, dwn, (unsigned char*)pool_argmax, parts);
  else if (dtype == EDET_F32) edet_launch(k_fuse<float, true>, grid, dim3(THREADS), lds, to_stream(stream), a, nullptr, (const float*)dout, (float*)ds, dwn, (unsigned char*)pool_argmax, parts);
  else EDET_CHECK(false, "edet_fuse_bwd_pre: bad dtype %d", dtype);
  if (parts) {
    FuseWBwd wb;
    memset(&wb, 0, sizeof(wb));
    if (wraw && dwraw && method != 1) {       // + the gradient of the raw variables (no edet_fuse_weights_bwd call needed)
      EDET_CHECK(method == 0 || method == 2, "edet_fuse_bwd_pre: bad method %d", method);
      for (int i = 0; i < nin; ++i) {
        EDET_CHECK(wraw[i] && dwraw[i], "edet_fuse_bwd_pre: null fusion variable %d", i);
        wb.w[i] = wraw[i];
        wb.dw[i] = dwraw[i];
      }
      wb.nin = nin;
      wb.method = method;
    }
    edet_launch(k_fuse_dwn_finish, dim3(1), dim3(256), 0, to_stream(stream), (const float*)parts, grid, dwn, wb);
  } else {
    EDET_CHECK(!(wraw && dwraw) || method == 1, "edet_fuse_bwd_pre: the fusion variables' gradient needs wc == 1 and a workspace of %zu bytes",
               (size_t)grid * 4 * sizeof(float));
  }
  EDET_LAUNCH_CHECK("edet_fuse_bwd_pre");
  return 0;
}

extern "C" int edet_fuse_bwd_input(const edet_tview_t* in, int mode, const float* wn, int wc, int idx,
                                   const void* ds, int oh, int ow, int lds_, const void* pool_argmax,
                                   void* gout, int beta, int dtype, void* stream) {
  EDET_CHECK(in && in->data && wn && ds && gout && idx >= 0 && idx < 3, "edet_fuse_bwd_input: bad arguments");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && lds_ % 8 == 0, "edet_fuse_bwd_input: c/ld % 8");
  EDET_CHECK(wc == 1 || wc == in->c, "edet_fuse_bwd_input: wc must be 1 or the channel count (got %d)", wc);
  FuseInArgs a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.mode = mode; a.wn = wn; a.wc = wc; a.idx = idx; a.n = in->n; a.oh = oh; a.ow = ow; a.ldds = lds_;
  a.beta = beta;
  if (mode == EDET_RS_POOL) {
    a.pad_t = same_pad_before(in->h, 3, 2);
    a.pad_l = same_pad_before(in->w, 3, 2);
  }
  EDET_CHECK((int64_t)in->n * in->h * in->w * (in->c / 8) < (1ll << 31), "edet_fuse_bwd_input: tensor too large for the 32-bit index arithmetic");
  const int grid = ew_grid((int64_t)in->n * in->h * in->w * (in->c / 8), dtype == EDET_BF16 ? reinterpret_cast<const void*>(&k_fuse_bwd_input<bf16_t>) : nullptr);
  const unsigned char* am = mode == EDET_RS_POOL ? reinterpret_cast<const unsigned char*>(pool_argmax) : nullptr;
  if (dtype == EDET_BF16) edet_launch(k_fuse_bwd_input<bf16_t>, grid, dim3(THREADS), 0, to_stream(stream), a, (const bf16_t*)ds, (bf16_t*)gout, am);
  else if (dtype == EDET_F32) edet_launch(k_fuse_bwd_input<float>, grid, dim3(THREADS), 0, to_stream(stream), a, (const float*)ds, (float*)gout, am);
  else EDET_CHECK(false, "edet_fuse_bwd_input: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_fuse_bwd_input");
  return 0;
}
