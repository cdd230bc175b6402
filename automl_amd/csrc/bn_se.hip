// BatchNorm statistics (forward finalize / backward reduce + finalize), block-output
// materialisation, squeeze-and-excitation, and small elementwise helpers.
//
// Reference: efficientdet/utils.py:166-266 (BatchNorm classes, eps 1e-3, momentum 0.99),
// efficientdet/tf2/util_keras.py:29-66, efficientdet/backbone/efficientnet_model.py:153-195 (SE),
// :393-410 (project BN + identity skip).
#include <algorithm>
#include <map>
#include <vector>

#include "common.h"

namespace {

constexpr int THREADS = 256;

// Row x channel-vector mapping with a *fixed* channel vector per thread, so per-channel partial
// sums can live in registers: tpr = threads per row (power of two >= c/8, <= 256).
struct RowMap {
  int tpr, rpp;  // threads per row, rows per pass
};
inline RowMap row_map(int c) {
  int nvec = c / 8, tpr = 1;
  while (tpr < nvec && tpr < THREADS) tpr <<= 1;
  RowMap m;
  m.tpr = tpr;
  m.rpp = THREADS / tpr;
  return m;
}
inline int persistent_grid(int64_t rows, int rpp, int max_wg) {
  int64_t passes = (rows + rpp - 1) / rpp;
  int64_t g = (passes + 7) / 8;  // at least ~8 passes per workgroup
  if (g < 1) g = 1;
  if (g > max_wg) g = max_wg;
  return (int)g;
}

// ------------------------------------------------------------------ forward statistics finalize
// Column sums of the [nparts][2][c] partial rows: one 1024-lane workgroup per FIN_CH channels, FIN_SL row slices per
// channel (8 rows = 16 independent loads in flight per thread), fp64 accumulation, the slices added in order at the end.
// r05 lab switches (compile time, the kernel symbols stay the same): EDET_FIN_SL = row slices per channel of the
// 1024-lane workgroup (32: 32 channels x 32 slices; 64: 16 channels x 64 slices -- half the rows per thread, twice the
// workgroups), EDET_FIN_DEEP = 1: a first tier with 16 rows (32 loads) in flight per thread.
#ifndef EDET_FIN_SL
#define EDET_FIN_SL 64      // r05d, same box, 30 steps: 32 -> 50.99 / 50.87 ms, 64 -> 50.71 ms, 64 + DEEP 50.97, 32 + DEEP 51.81
#endif
#ifndef EDET_FIN_DEEP
#define EDET_FIN_DEEP 0
#endif
constexpr int FIN_SL = EDET_FIN_SL, FIN_CH = 1024 / FIN_SL;   // 1024-lane workgroups: FIN_SL rows of partials per step

__device__ __forceinline__ void partial_colsum(const float* __restrict__ partials, int nparts, int c, int ch,
                                               int slice, double& s, double& s2) {
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (ch < c) {
    int p = slice;
#if EDET_FIN_DEEP
    for (; p + 15 * FIN_SL < nparts; p += 16 * FIN_SL) {      // sixteen rows in flight, same order of additions
      float u[16], v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        u[i] = partials[((size_t)(p + i * FIN_SL) * 2) * c + ch];
        v[i] = partials[((size_t)(p + i * FIN_SL) * 2 + 1) * c + ch];
      }
#pragma unroll
      for (int i = 0; i < 16; i += 2) { a0 += (double)u[i]; b0 += (double)v[i]; a1 += (double)u[i + 1]; b1 += (double)v[i + 1]; }
    }
#endif
    // eight rows (16 loads) in flight per thread first -- the kernel is a chain of L2 latencies, ~10 us of step time per
    // launch and 216 launches per EfficientDet-D0 step -- added in exactly the order of the two-row loop below (same bits)
    for (; p + 7 * FIN_SL < nparts; p += 8 * FIN_SL) {
      float u[8], v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u[i] = partials[((size_t)(p + i * FIN_SL) * 2) * c + ch];
        v[i] = partials[((size_t)(p + i * FIN_SL) * 2 + 1) * c + ch];
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2) { a0 += (double)u[i]; b0 += (double)v[i]; a1 += (double)u[i + 1]; b1 += (double)v[i + 1]; }
    }
    for (; p + FIN_SL < nparts; p += 2 * FIN_SL) {
      const float u0 = partials[((size_t)p * 2) * c + ch], v0 = partials[((size_t)p * 2 + 1) * c + ch];
      const float u1 = partials[((size_t)(p + FIN_SL) * 2) * c + ch];
      const float v1 = partials[((size_t)(p + FIN_SL) * 2 + 1) * c + ch];
      a0 += (double)u0; b0 += (double)v0; a1 += (double)u1; b1 += (double)v1;
    }
    if (p < nparts) {
      a0 += (double)partials[((size_t)p * 2) * c + ch];
      b0 += (double)partials[((size_t)p * 2 + 1) * c + ch];
    }
  }
  __shared__ double red[2][FIN_SL][FIN_CH];
  const int cl = threadIdx.x & (FIN_CH - 1);
  red[0][slice][cl] = a0 + a1;
  red[1][slice][cl] = b0 + b1;
  __syncthreads();
  // the slices in two fixed levels (r06: one thread adding all FIN_SL slices was a chain of FIN_SL dependent LDS reads,
  // ~1 us of a ~6 us kernel that runs 216 times per step): FIN_G leaders add FIN_SL / FIN_G consecutive slices each,
  // slice 0 adds the leaders in order -- the same tree on every run
  constexpr int FIN_G = 8, PER = FIN_SL / FIN_G;
  static_assert(FIN_SL % FIN_G == 0, "slice groups");
  double u = 0.0, v = 0.0;
  if (slice < FIN_G) {
#pragma unroll
    for (int i = 0; i < PER; ++i) { u += red[0][slice * PER + i][cl]; v += red[1][slice * PER + i][cl]; }
  }
  __syncthreads();                   // every leader has read its slices before any of them overwrites rows 0 .. FIN_G - 1
  if (slice < FIN_G) {
    red[0][slice][cl] = u;
    red[1][slice][cl] = v;
  }
  __syncthreads();
  s = 0.0; s2 = 0.0;
  if (slice == 0) {
#pragma unroll
    for (int i = 0; i < FIN_G; ++i) { s += red[0][i][cl]; s2 += red[1][i][cl]; }
  }
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void k_bn_finalize(
    const float* __restrict__ partials, int nparts, int c, double count, const float* gamma, const float* beta,
    float eps, float momentum, int bessel, float* moving_mean, float* moving_var, float* scale, float* shift,
    float* mean_out, float* rstd_out) {
  const int ch = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  const int slice = threadIdx.x / FIN_CH;
  double s, s2;
  partial_colsum(partials, nparts, c, ch, slice, s, s2);
  if (slice != 0 || ch >= c) return;
  const double mean = s / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[ch] * rstd;
  scale[ch] = sc;
  shift[ch] = beta[ch] - (float)mean * sc;
  mean_out[ch] = (float)mean;
  rstd_out[ch] = rstd;
  if (momentum >= 0.f && moving_mean) {
    // Keras fused BatchNorm (the single-replica classes): moving variance is updated with the Bessel-corrected batch
    // variance; SyncBatchNormalization / TpuBatchNormalization run un-fused (utils.py:172,211): biased variance
    const double unbiased = (bessel && count > 1.0) ? var * count / (count - 1.0) : var;
    moving_mean[ch] = moving_mean[ch] * momentum + (float)mean * (1.f - momentum);
    moving_var[ch] = moving_var[ch] * momentum + (float)unbiased * (1.f - momentum);
  }
}

__global__ void k_bn_eval(int c, const float* gamma, const float* beta, float eps,
                          const float* moving_mean, const float* moving_var, float* scale, float* shift) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float sc = gamma[ch] * rsqrtf(moving_var[ch] + eps);
  scale[ch] = sc;
  shift[ch] = beta[ch] - moving_mean[ch] * sc;
}

// ------------------------------------------------------------------ backward reduce / finalize
// The per-channel sums of a workgroup WITHOUT LDS atomics (r04: run-to-run identical results): the row-lanes (rr) of a
// channel chunk park their 8 + 8 sums in scr[2][THREADS * 8] and the first tpr * 8 threads add them in row-lane order
// into red[cb ..] / red[c + cb ..].  Called by every thread of the workgroup (two barriers).
__device__ __forceinline__ void rowlane_sums(float* scr, const RowMap& m, int cv, int rr, bool ok, const float (&s1)[8],
                                             const float (&s2)[8], float* red, int cb, int c) {
  const int width = m.tpr * 8;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    scr[rr * width + cv * 8 + e] = ok ? s1[e] : 0.f;
    scr[THREADS * 8 + rr * width + cv * 8 + e] = ok ? s2[e] : 0.f;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < width; t += THREADS) {
    float u = 0.f, v = 0.f;
    for (int r = 0; r < m.rpp; ++r) { u += scr[r * width + t]; v += scr[THREADS * 8 + r * width + t]; }
    if (cb + t < c) { red[cb + t] = u; red[c + cb + t] = v; }
  }
}

template <typename T>
__global__ __launch_bounds__(THREADS) void k_bn_bwd_reduce(const T* __restrict__ dz, const T* __restrict__ y,
                                                          int64_t rows, int c, int ld, const float* mean,
                                                          const float* rstd, float* partials, RowMap m) {
  const int tid = threadIdx.x;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  extern __shared__ float red[];  // [2][c] + scratch [2][THREADS * 8]
  float* scr = red + 2 * c;
  // channel blocks of tpr*8 (<= 2048) channels: one iteration for every layer up to 2048 channels, two for the
  // widest EfficientNet-B3..B7 stages
  for (int cb = 0; cb < c; cb += m.tpr * 8) {
    const int c0 = cb + cv * 8;
    const bool ok = c0 < c;
    float mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = 0.f; rs[e] = 0.f; s1[e] = s2[e] = 0.f; }
    if (ok) {
      loadf8(mean + c0, mu);
      loadf8(rstd + c0, rs);
      // three rows (six 16-byte loads) in flight per thread, see k_se_pool; sums in row order
      const int64_t st = (int64_t)gridDim.x * m.rpp;
      int64_t r = (int64_t)blockIdx.x * m.rpp + rr;
      for (; r + 2 * st < rows; r += 3 * st) {
        float g[3][8], x[3][8];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          load8<T>(dz + (r + u * st) * ld + c0, g[u]);
          load8<T>(y + (r + u * st) * ld + c0, x[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += g[u][e]; s2[e] += g[u][e] * (x[u][e] - mu[e]) * rs[e]; }
      }
      for (; r < rows; r += st) {
        float g[8], x[8];
        load8<T>(dz + r * ld + c0, g);
        load8<T>(y + r * ld + c0, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += g[e]; s2[e] += g[e] * (x[e] - mu[e]) * rs[e]; }
      }
    }
    rowlane_sums(scr, m, cv, rr, ok, s1, s2, red, cb, c);
  }
  __syncthreads();
  for (int i = tid; i < 2 * c; i += THREADS) partials[(size_t)blockIdx.x * 2 * c + i] = red[i];
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void k_bn_bwd_finalize(
    const float* __restrict__ partials, int nparts, int c, double count, const float* gamma, const float* mean,
    const float* rstd, float* dgamma, float* dbeta, float* a, float* b, float* cc) {
  const int ch = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  const int slice = threadIdx.x / FIN_CH;
  double s1, s2;
  partial_colsum(partials, nparts, c, ch, slice, s1, s2);
  if (slice != 0 || ch >= c) return;
  if (dbeta) dbeta[ch] += (float)s1;
  if (dgamma) dgamma[ch] += (float)s2;
  const double m1 = s1 / count, m2 = s2 / count;
  const double g = gamma[ch], r = rstd[ch], mu = mean[ch];
  // dy = g*r*(dz - m1 - xhat*m2), xhat = (y - mu)*r
  a[ch] = (float)(g * r);
  b[ch] = (float)(-g * r * r * m2);
  cc[ch] = (float)(-g * r * m1 + g * r * r * m2 * mu);
}

// ------------------------------------------------------------------ out = y*scale+shift (+res)
template <typename T>
__global__ void k_bn_res(const edet_tview_t y, const T* __restrict__ res, T* __restrict__ out, int ldo,
                         int64_t rows) {
  const int nvec = y.c / 8;
  const int64_t total = rows * nvec;
  const int hw = y.h * y.w;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = q / nvec;
    const int c0 = (int)(q - r * nvec) * 8;
    float x[8];
    load8<T>(reinterpret_cast<const T*>(y.data) + r * y.ld + c0, x);
    ViewCoef vc;
    view_load_coef(y, c0, vc);
    view_apply(y, vc, c0, y.gate ? (int)(r / hw) : 0, x);   // gate [n][c]: stochastic-depth scale per image
    if (res) {
      float rr[8];
      load8<T>(res + r * ldo + c0, rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += rr[e];
    }
    store8<T>(out + r * ldo + c0, x);
  }
}

template <typename T>
__global__ void k_add(T* __restrict__ dst, const T* __restrict__ src, int64_t rows, int c, int ld, int beta) {
  const int nvec = c / 8;
  const int64_t total = rows * nvec;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = q / nvec;
    const int c0 = (int)(q - r * nvec) * 8;
    float x[8];
    load8<T>(src + r * ld + c0, x);
    if (beta) {
      float d[8];
      load8<T>(dst + r * ld + c0, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += d[e];
    }
    store8<T>(dst + r * ld + c0, x);
  }
}

// ------------------------------------------------------------------ squeeze-and-excitation
// Global average pooling WITHOUT atomics and with a summation order that does not depend on the batch: an image's
// rows are cut into chunks of `cr` rows (a function of the map and channel count only, se_chunk_rows), one workgroup
// per (image, chunk); thread (rr, cv) adds the rows rr, rr + rpp, ... of its chunk in row order, the rpp row-slices of
// a workgroup are added in slice order through LDS, and the chunk sums of an image are added in chunk order by the
// consumer (k_se_pool_finish or k_se_fc).  The same image therefore gives bit-identical pooled sums whatever batch it
// sits in and from run to run (the first version used fp32 atomics on both levels).
// parts[(n * nchunks + chunk) * c + i] = sum over the chunk's pixels of view(in)
template <typename T>
__global__ __launch_bounds__(THREADS) void k_se_pool(const edet_tview_t in, float* __restrict__ parts, int nchunks,
                                                    int cr, RowMap m) {
  const int tid = threadIdx.x;
  const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  const int hw = in.h * in.w;
  const int r0 = chunk * cr, r1 = min(hw, r0 + cr);
  extern __shared__ float red[];  // [rpp][cblk], cblk = min(c, tpr * 8)
  const int cblk = min(in.c, m.tpr * 8);
  edet_tview_t v = in;
  v.gate = nullptr;
  const T* base = reinterpret_cast<const T*>(in.data) + (size_t)n * hw * in.ld;
  float* out = parts + ((size_t)n * nchunks + chunk) * in.c;
  for (int cb = 0; cb < in.c; cb += m.tpr * 8) {     // channel blocks of <= 2048 channels
    const int c0 = cb + cv * 8;
    if (c0 < in.c) {
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = 0.f;
      ViewCoef vc;
      view_load_coef(in, c0, vc);
      // four rows of loads in flight per thread (16 waves/CU x 16 B per lane is ~4 MB in flight over the chip, a
      // third of what 2 us of HBM latency at 5 TB/s needs); the sums keep their row order
      const int st = m.rpp;
      int r = r0 + rr;
      for (; r + 3 * st < r1; r += 4 * st) {
        float x[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<T>(base + (size_t)(r + u * st) * in.ld + c0, x[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          view_apply(v, vc, c0, n, x[u]);
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += x[u][e];
        }
      }
      for (; r < r1; r += st) {
        float x[8];
        load8<T>(base + (size_t)r * in.ld + c0, x);
        view_apply(v, vc, c0, n, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += x[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rr * cblk + cv * 8 + e] = s[e];
    }
    __syncthreads();
    for (int i = tid; i < cblk && cb + i < in.c; i += THREADS) {
      float t = red[i];
      for (int q = 1; q < m.rpp; ++q) t += red[q * cblk + i];      // row slices in slice order
      out[cb + i] = t;
    }
    __syncthreads();
  }
}

// pooled[n][i] = chunk sums of image n added in chunk order
__global__ __launch_bounds__(THREADS) void k_se_pool_finish(const float* __restrict__ parts, int nchunks, int c,
                                                           float* __restrict__ pooled, int total) {
  const int idx = blockIdx.x * THREADS + threadIdx.x;
  if (idx >= total) return;
  const int n = idx / c, i = idx - n * c;
  const float* p = parts + (size_t)n * nchunks * c + i;
  float t = p[0];
  for (int k = 1; k < nchunks; ++k) t += p[(size_t)k * c];
  pooled[idx] = t;
}

// one workgroup (SE_FC_THREADS lanes) per image.  pooled_parts != nullptr: the pooled sums are still k_se_pool's chunk
// rows [n][nchunks][c]; they are added here in chunk order and written to pooled (the backward pass reads them).
constexpr int SE_FC_THREADS = 1024;
__global__ __launch_bounds__(SE_FC_THREADS) void k_se_fc(float* __restrict__ pooled,
                                                        const float* __restrict__ pooled_parts, int nchunks, int c,
                                                        int se, float inv_hw, const float* w1, const float* b1,
                                                        const float* w2, const float* b2, float* hidden_pre,
                                                        float* gate, int act) {
  extern __shared__ float sm[];  // p[c], h[se], hp[nthr]
  float* p = sm;
  float* h = sm + c;
  float* hp = sm + c + se;
  const int n = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < c; i += nthr) {
    float t;
    if (pooled_parts) {
      const float* q = pooled_parts + (size_t)n * nchunks * c + i;
      t = q[0];
      for (int k = 1; k < nchunks; ++k) t += q[(size_t)k * c];
      pooled[(size_t)n * c + i] = t;
    } else {
      t = pooled[(size_t)n * c + i];
    }
    p[i] = t * inv_hw;
  }
  __syncthreads();
  // thread (j, part): hidden unit j over the channels i = part, part + nparts, ... (w1 loads coalesced along
  // j; four independent partial sums keep four loads in flight); the parts of a unit are added in part order
  if (se <= nthr) {
    const int nparts = nthr / se, j = tid % se, part = tid / se;
    if (part < nparts) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int i = part;
      for (; i + 3 * nparts < c; i += 4 * nparts) {
        a0 = fmaf(p[i], w1[(size_t)i * se + j], a0);
        a1 = fmaf(p[i + nparts], w1[(size_t)(i + nparts) * se + j], a1);
        a2 = fmaf(p[i + 2 * nparts], w1[(size_t)(i + 2 * nparts) * se + j], a2);
        a3 = fmaf(p[i + 3 * nparts], w1[(size_t)(i + 3 * nparts) * se + j], a3);
      }
      for (; i < c; i += nparts) a0 = fmaf(p[i], w1[(size_t)i * se + j], a0);
      hp[part * se + j] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (tid < se) {
      float t = hp[tid];
      for (int q = 1; q < nparts; ++q) t += hp[q * se + tid];
      h[tid] = t;
    }
  } else {
    for (int j = tid; j < se; j += nthr) {
      float acc = 0.f;
      for (int i = 0; i < c; ++i) acc = fmaf(p[i], w1[(size_t)i * se + j], acc);
      h[j] = acc;
    }
  }
  __syncthreads();
  for (int j = tid; j < se; j += nthr) {
    const float acc = h[j] + b1[j];
    hidden_pre[(size_t)n * se + j] = acc;
    h[j] = act_apply_(act, acc);
  }
  __syncthreads();
  for (int i = tid; i < c; i += nthr) {
    float a0 = b2[i], a1 = 0.f;
    int j = 0;
    for (; j + 1 < se; j += 2) {
      a0 = fmaf(h[j], w2[(size_t)j * c + i], a0);
      a1 = fmaf(h[j + 1], w2[(size_t)(j + 1) * c + i], a1);
    }
    if (j < se) a0 = fmaf(h[j], w2[(size_t)j * c + i], a0);
    gate[(size_t)n * c + i] = sigmoidf_(a0 + a1);
  }
}

// ---- the two 1x1 layers for WIDE squeeze-and-excitation blocks (c * se >= SE_SPLIT_MIN) ------------------------------
// k_se_fc streams both weight matrices through ONE compute unit per image: 3840 x 160 is 4.9 MB (144 us per call at
// efficientdet-d7x batch 8, 7.5 ms per step), and at batch 256 (efficientnetv2-s, 1536 x 64) every image's workgroup
// re-reads the same 0.8 MB from L2 (17 us per call, L2-bound).  Here the channel axis is cut into slices of SE_SLICE
// channels -- a function of c only, so the summation order of an image does not depend on the batch -- and a workgroup
// handles SE_IB images at once, so a weight element is loaded once per SE_IB images: (slice, image block) workgroups
// produce partial hidden sums (k_se_fc1_split), (slice, image block) workgroups add them in slice order, apply bias +
// activation and compute the gates of their slice (k_se_fc2_split).  Each image's sums are formed exactly as with
// SE_IB = 1 (its own accumulators, same order).
constexpr int SE_SLICE = 128;
constexpr int SE_IB = 4;
constexpr int SE_SPLIT_MIN = 1 << 15;       // forward (image-blocked: also pays at large batch)
constexpr int SE_SPLIT_MIN_BWD = 1 << 17;   // backward (per image: pays where one CU per image is the bottleneck)
__global__ __launch_bounds__(THREADS) void k_se_fc1_split(float* __restrict__ pooled,
                                                         const float* __restrict__ pooled_parts, int nchunks, int c,
                                                         int se, float inv_hw, const float* __restrict__ w1,
                                                         float* __restrict__ hpart, int nslice, int nimg) {
  __shared__ float p[SE_IB][SE_SLICE];
  __shared__ float hp[SE_IB][THREADS];
  const int sl = blockIdx.x, n0 = blockIdx.y * SE_IB, tid = threadIdx.x;
  const int c0 = sl * SE_SLICE, cn = min(SE_SLICE, c - c0);
  for (int q = tid; q < SE_IB * SE_SLICE; q += THREADS) {
    const int b = q / SE_SLICE, i = q - b * SE_SLICE, n = n0 + b;
    float t = 0.f;
    if (i < cn && n < nimg) {
      if (pooled_parts) {
        const float* src = pooled_parts + (size_t)n * nchunks * c + c0 + i;
        t = src[0];
        for (int k = 1; k < nchunks; ++k) t += src[(size_t)k * c];
        pooled[(size_t)n * c + c0 + i] = t;
      } else {
        t = pooled[(size_t)n * c + c0 + i];
      }
    }
    p[b][i] = t * inv_hw;
  }
  __syncthreads();
  // thread (j, part): hidden unit j over the slice's channels part, part + nparts, ...; parts added in part order
  for (int j0 = 0; j0 < se; j0 += THREADS) {           // se <= THREADS in practice: one round
    const int seb = min(THREADS, se - j0);
    const int nparts = THREADS / seb, j = tid % seb, part = tid / seb;
    if (part < nparts) {
      float a0[SE_IB], a1[SE_IB];
#pragma unroll
      for (int b = 0; b < SE_IB; ++b) a0[b] = a1[b] = 0.f;
      int i = part;
      for (; i + nparts < cn; i += 2 * nparts) {
        const float u = w1[(size_t)(c0 + i) * se + j0 + j], v = w1[(size_t)(c0 + i + nparts) * se + j0 + j];
#pragma unroll
        for (int b = 0; b < SE_IB; ++b) { a0[b] = fmaf(p[b][i], u, a0[b]); a1[b] = fmaf(p[b][i + nparts], v, a1[b]); }
      }
      if (i < cn) {
        const float u = w1[(size_t)(c0 + i) * se + j0 + j];
#pragma unroll
        for (int b = 0; b < SE_IB; ++b) a0[b] = fmaf(p[b][i], u, a0[b]);
      }
#pragma unroll
      for (int b = 0; b < SE_IB; ++b) hp[b][part * seb + j] = a0[b] + a1[b];
    }
    __syncthreads();
    for (int q = tid; q < SE_IB * seb; q += THREADS) {
      const int b = q / seb, jj = q - b * seb;
      if (n0 + b < nimg) {
        float t = hp[b][jj];
        for (int r = 1; r < nparts; ++r) t += hp[b][r * seb + jj];
        hpart[((size_t)(n0 + b) * nslice + sl) * se + j0 + jj] = t;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(THREADS) void k_se_fc2_split(const float* __restrict__ hpart, int nslice, int c, int se,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ hidden_pre,
                                                         float* __restrict__ gate, int act, int nimg) {
  extern __shared__ float h[];       // [SE_IB][se]
  const int sl = blockIdx.x, n0 = blockIdx.y * SE_IB, tid = threadIdx.x;
  for (int q = tid; q < SE_IB * se; q += THREADS) {
    const int b = q / se, j = q - b * se, n = n0 + b;
    float v = 0.f;
    if (n < nimg) {
      const float* src = hpart + (size_t)n * nslice * se + j;
      float t = src[0];
      for (int k = 1; k < nslice; ++k) t += src[(size_t)k * se];       // slices in slice order
      const float acc = t + b1[j];
      if (sl == 0) hidden_pre[(size_t)n * se + j] = acc;
      v = act_apply_(act, acc);
    }
    h[q] = v;
  }
  __syncthreads();
  const int c0 = sl * SE_SLICE;
  for (int i = c0 + tid; i < min(c, c0 + SE_SLICE); i += THREADS) {
    float a0[SE_IB], a1[SE_IB];
#pragma unroll
    for (int b = 0; b < SE_IB; ++b) { a0[b] = b2[i]; a1[b] = 0.f; }
    int j = 0;
    for (; j + 1 < se; j += 2) {
      const float u = w2[(size_t)j * c + i], v = w2[(size_t)(j + 1) * c + i];
#pragma unroll
      for (int b = 0; b < SE_IB; ++b) { a0[b] = fmaf(h[b * se + j], u, a0[b]); a1[b] = fmaf(h[b * se + j + 1], v, a1[b]); }
    }
    if (j < se) {
      const float u = w2[(size_t)j * c + i];
#pragma unroll
      for (int b = 0; b < SE_IB; ++b) a0[b] = fmaf(h[b * se + j], u, a0[b]);
    }
#pragma unroll
    for (int b = 0; b < SE_IB; ++b)
      if (n0 + b < nimg) gate[(size_t)(n0 + b) * c + i] = sigmoidf_(a0[b] + a1[b]);
  }
}

// per image: dgate -> dpre2, dh, dpre1, dpool.
// scratch: dpre2 [n][c], dpre1 [n][se], hact = swish(hidden_pre) [n][se]
__global__ __launch_bounds__(SE_FC_THREADS) void k_se_fc_bwd_img(const float* __restrict__ hidden_pre,
                                                          const float* __restrict__ gate,
                                                          const float* __restrict__ dgate, int nimg, int c,
                                                          int se, float inv_hw, const float* w1,
                                                          const float* w2, float* dpool, float* scratch, int act) {
  extern __shared__ float sm[];  // dpre2[c], dpre1[se]
  float* d2 = sm;
  float* d1 = sm + c;
  const int n = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  float* dpre2_g = scratch + (size_t)n * c;
  float* dpre1_g = scratch + (size_t)nimg * c + (size_t)n * se;
  float* hact_g = scratch + (size_t)nimg * (c + se) + (size_t)n * se;
  for (int i = tid; i < c; i += nthr) {
    const float g = gate[(size_t)n * c + i];
    const float v = dgate[(size_t)n * c + i] * g * (1.f - g);
    d2[i] = v;
    dpre2_g[i] = v;
  }
  __syncthreads();
  // dh[j] = sum_i dpre2[i] * w2[j][i]: one wave per j (coalesced along i), shuffle reduction
  const int wave = tid >> 6, lane = tid & 63;
  for (int j = wave; j < se; j += nthr / 64) {
    float acc = 0.f;
    for (int i = lane; i < c; i += 64) acc = fmaf(d2[i], w2[(size_t)j * c + i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
      const float hp = hidden_pre[(size_t)n * se + j];
      const float v = acc * act_grad_(act, hp);
      d1[j] = v;
      dpre1_g[j] = v;
      hact_g[j] = act_apply_(act, hp);
    }
  }
  __syncthreads();
  for (int i = tid; i < c; i += nthr) {
    float acc = 0.f;
    for (int j = 0; j < se; ++j) acc = fmaf(d1[j], w1[(size_t)i * se + j], acc);
    dpool[(size_t)n * c + i] = acc * inv_hw;
  }
}

// the same per-image work for WIDE blocks (c * se >= SE_SPLIT_MIN_BWD), sliced over the channel axis like k_se_fc1_split /
// k_se_fc2_split: (slice, image) workgroups make dpre2 and the slice's share of dh, then add the shares in slice order,
// finish dpre1 / the activated hidden units and compute dpool of their slice.  hpart: [n][nslice][se] behind the
// caller's scratch rows.
__global__ __launch_bounds__(THREADS) void k_se_fc_bwd_img1(const float* __restrict__ gate,
                                                           const float* __restrict__ dgate, int nimg, int c, int se,
                                                           const float* __restrict__ w2, float* __restrict__ scratch,
                                                           float* __restrict__ hpart, int nslice) {
  __shared__ float d2[SE_SLICE];
  const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int c0 = sl * SE_SLICE, cn = min(SE_SLICE, c - c0);
  float* dpre2_g = scratch + (size_t)n * c;
  for (int i = tid; i < SE_SLICE; i += THREADS) {
    float v = 0.f;
    if (i < cn) {
      const float g = gate[(size_t)n * c + c0 + i];
      v = dgate[(size_t)n * c + c0 + i] * g * (1.f - g);
      dpre2_g[c0 + i] = v;
    }
    d2[i] = v;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  float* dst = hpart + ((size_t)n * nslice + sl) * se;
  for (int j = wave; j < se; j += THREADS / 64) {
    float acc = 0.f;
    for (int i = lane; i < cn; i += 64) acc = fmaf(d2[i], w2[(size_t)j * c + c0 + i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) dst[j] = acc;
  }
}

__global__ __launch_bounds__(THREADS) void k_se_fc_bwd_img2(const float* __restrict__ hidden_pre,
                                                           const float* __restrict__ hpart, int nslice, int nimg,
                                                           int c, int se, float inv_hw, const float* __restrict__ w1,
                                                           float* __restrict__ dpool, float* __restrict__ scratch,
                                                           int act) {
  extern __shared__ float d1[];      // [se]
  const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  float* dpre1_g = scratch + (size_t)nimg * c + (size_t)n * se;
  float* hact_g = scratch + (size_t)nimg * (c + se) + (size_t)n * se;
  for (int j = tid; j < se; j += THREADS) {
    const float* q = hpart + (size_t)n * nslice * se + j;
    float t = q[0];
    for (int k = 1; k < nslice; ++k) t += q[(size_t)k * se];
    const float hp = hidden_pre[(size_t)n * se + j];
    const float v = t * act_grad_(act, hp);
    d1[j] = v;
    if (sl == 0) {
      dpre1_g[j] = v;
      hact_g[j] = act_apply_(act, hp);
    }
  }
  __syncthreads();
  const int c0 = sl * SE_SLICE;
  for (int i = c0 + tid; i < min(c, c0 + SE_SLICE); i += THREADS) {
    float acc = 0.f;
    for (int j = 0; j < se; ++j) acc = fmaf(d1[j], w1[(size_t)i * se + j], acc);
    dpool[(size_t)n * c + i] = acc * inv_hw;
  }
}

// parameter gradients.  Workgroup = 64 channels i x 4 hidden-unit groups, for one slice of the images
// (blockIdx.y) and one block of SE_JB hidden units (blockIdx.z); thread (i, jg) owns the (i, j) pairs with
// j % 4 == jg and sums over its images (loads coalesced along i).  r04: every image slice writes its sums to its own
// partial [dw1 | dw2 | db1 | db2] in the scratch and k_se_fc_bwd_sum adds the SE_SPLIT partials in slice order -- the
// same gradients on every run (rounds 2-3 combined the slices with fp32 atomics; summing all images in one workgroup
// instead was measured at twice the time: 45 -> 89 us per call over the 16 blocks of D0).
//   dw1[i][j] += sum_n pooled[n][i]*inv_hw * dpre1[n][j];  dw2[j][i] += sum_n hact[n][j] * dpre2[n][i]
constexpr int SE_MAX_JPT = 12;           // hidden units per thread
constexpr int SE_JB = 4 * SE_MAX_JPT;    // hidden units per workgroup (48)
constexpr int SE_NB = 64;                // images per LDS chunk
constexpr int SE_SPLIT = 8;              // image slices
__global__ __launch_bounds__(THREADS) void k_se_fc_bwd_par(const float* __restrict__ pooled,
                                                          const float* __restrict__ scratch, int nimg, int c,
                                                          int se, float inv_hw, float* __restrict__ parts,
                                                          int per_split) {
  extern __shared__ float sm[];  // dpre1 [NB][jb], hact [NB][jb] for the current chunk of images
  const float* dpre2 = scratch;
  const float* dpre1_g = scratch + (size_t)nimg * c;
  const float* hact_g = scratch + (size_t)nimg * (c + se);
  const int j0 = blockIdx.z * SE_JB, jb = min(SE_JB, se - j0);
  float* d1 = sm;
  float* ha = sm + (size_t)SE_NB * SE_JB;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 64 + (tid & 63), jg = tid >> 6;
  const int nbeg = blockIdx.y * per_split, nend = min(nimg, nbeg + per_split);
  const size_t cs = (size_t)c * se;
  float* pw1 = parts + (size_t)blockIdx.y * (2 * cs + se + c);      // this slice's partial: dw1 [c][se]
  float* pw2 = pw1 + cs;                                             // dw2 [se][c]
  float* pb1 = pw2 + cs;                                             // db1 [se]
  float* pb2 = pb1 + se;                                             // db2 [c]
  float a1[SE_MAX_JPT], a2[SE_MAX_JPT];
#pragma unroll
  for (int t = 0; t < SE_MAX_JPT; ++t) a1[t] = a2[t] = 0.f;
  float sb2 = 0.f, sb1 = 0.f;
  for (int n0 = nbeg; n0 < nend; n0 += SE_NB) {
    const int nb = min(SE_NB, nend - n0);
    __syncthreads();
    for (int q = tid; q < nb * jb; q += THREADS) {
      const int n = q / jb, j = q - n * jb;
      d1[q] = dpre1_g[(size_t)(n0 + n) * se + j0 + j];
      ha[q] = hact_g[(size_t)(n0 + n) * se + j0 + j];
    }
    __syncthreads();
    if (i < c) {
      for (int n = 0; n < nb; ++n) {
        const float pl = pooled[(size_t)(n0 + n) * c + i] * inv_hw;
        const float d2 = dpre2[(size_t)(n0 + n) * c + i];
        sb2 += d2;
#pragma unroll
        for (int t = 0; t < SE_MAX_JPT; ++t) {
          const int j = jg + 4 * t;
          if (j < jb) {
            a1[t] = fmaf(pl, d1[n * jb + j], a1[t]);
            a2[t] = fmaf(ha[n * jb + j], d2, a2[t]);
          }
        }
      }
    }
    if (blockIdx.x == 0 && tid < jb)
      for (int n = 0; n < nb; ++n) sb1 += d1[n * jb + tid];
  }
  if (blockIdx.x == 0 && tid < jb) pb1[j0 + tid] = sb1;
  if (i < c) {
    if (jg == 0 && blockIdx.z == 0) pb2[i] = sb2;
#pragma unroll
    for (int t = 0; t < SE_MAX_JPT; ++t) {
      const int j = jg + 4 * t;
      if (j < jb) {
        pw1[(size_t)i * se + j0 + j] = a1[t];
        pw2[(size_t)(j0 + j) * c + i] = a2[t];
      }
    }
  }
}

// (dw1, dw2, db1, db2) += the nsplit partials, in slice order
__global__ __launch_bounds__(THREADS) void k_se_fc_bwd_sum(const float* __restrict__ parts, int nsplit, int c, int se,
                                                          float* dw1, float* db1, float* dw2, float* db2) {
  const size_t cs = (size_t)c * se, total = 2 * cs + se + c;
  const size_t q = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (q >= total) return;
  float t = 0.f;
  for (int p = 0; p < nsplit; ++p) t += parts[(size_t)p * total + q];
  if (q < cs) dw1[q] += t;
  else if (q < 2 * cs) dw2[q - cs] += t;
  else if (q < 2 * cs + se) db1[q - 2 * cs] += t;
  else db2[q - 2 * cs - se] += t;
}

// g (in place, holds D) -> dz = (D*gate + dpool)*act'(z); BN backward stat partials.  OTHER: an activation beyond
// swish (its own instantiation: the extra selects cost the swish kernel three VGPRs and one occupancy step)
template <typename T, bool OTHER>
__global__ __launch_bounds__(THREADS) void k_se_gate_bwd(const edet_tview_t in, T* g, const float* dpool,
                                                        const float* mean, const float* rstd,
                                                        float* partials, int wg_per_img, RowMap m) {
  const int tid = threadIdx.x;
  const int n = blockIdx.x / wg_per_img, part = blockIdx.x % wg_per_img;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  const int hw = in.h * in.w;
  extern __shared__ float red[];  // [2][c] + scratch [2][THREADS * 8]
  float* scr = red + 2 * in.c;
  for (int cb = 0; cb < in.c; cb += m.tpr * 8) {     // channel blocks of <= 2048 channels
  const int c0 = cb + cv * 8;
  const bool ok = c0 < in.c;
  float s1[8], s2[8], sc[8], sh[8], mu[8], rs[8], gt[8], dp[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; gt[e] = 1.f; dp[e] = 0.f; }
  if (ok) {
    if (in.scale) { loadf8(in.scale + c0, sc); loadf8(in.shift + c0, sh); }
    loadf8(mean + c0, mu);
    loadf8(rstd + c0, rs);
    loadf8(in.gate + (size_t)n * in.c + c0, gt);
    loadf8(dpool + (size_t)n * in.c + c0, dp);
    const size_t base = (size_t)n * hw * in.ld;
    auto chain = [&](float (&d)[8], const float (&x)[8]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = fmaf(x[e], sc[e], sh[e]);
        const float da = fmaf(d[e], gt[e], dp[e]);
        if (OTHER) d[e] = da * act_other_grad_(in.act, z);
        else d[e] = in.act == EDET_ACT_SWISH ? da * swish_gradf_(z) : da;
        s1[e] += d[e];
        s2[e] += d[e] * (x[e] - mu[e]) * rs[e];
      }
    };
    // two rows (four 16-byte loads) in flight per thread, see k_se_pool (three rows would cost the fourth wave per
    // SIMD that the 1024-workgroup grid needs to be resident at once); sums in row order
    const int st = wg_per_img * m.rpp;
    int r = part * m.rpp + rr;
    for (; r + st < hw; r += 2 * st) {
      float d[2][8], x[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const size_t off = base + (size_t)(r + u * st) * in.ld + c0;
        load8<T>(g + off, d[u]);
        load8<T>(reinterpret_cast<const T*>(in.data) + off, x[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        chain(d[u], x[u]);
        store8<T>(g + base + (size_t)(r + u * st) * in.ld + c0, d[u]);
      }
    }
    for (; r < hw; r += st) {
      const size_t off = base + (size_t)r * in.ld + c0;
      float d[8], x[8];
      load8<T>(g + off, d);
      load8<T>(reinterpret_cast<const T*>(in.data) + off, x);
      chain(d, x);
      store8<T>(g + off, d);
    }
  }
  rowlane_sums(scr, m, cv, rr, ok, s1, s2, red, cb, in.c);
  }
  __syncthreads();
  for (int i = tid; i < 2 * in.c; i += THREADS) partials[(size_t)blockIdx.x * 2 * in.c + i] = red[i];
}

// dst[i] += sum over the P partial rows: thread (element e = tid & 15, slice sl = tid >> 4) sums rows
// sl, sl+64, ... with two independent loads in flight, the 64 slices are combined through LDS.
constexpr int RED_SL = 64;      // row slices per element: 16 elements x 64 slices = 1024 lanes
__global__ __launch_bounds__(16 * RED_SL) void k_reduce_partials(const float* __restrict__ ws, int P, int64_t n,
                                                                float* __restrict__ dst, int accumulate,
                                                                int64_t n_a = -1, float* __restrict__ dst_b = nullptr) {
  __shared__ float red[RED_SL][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + e;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int p = sl;
    for (; p + RED_SL < P; p += 2 * RED_SL) {
      s0 += ws[(size_t)p * n + i];
      s1 += ws[(size_t)(p + RED_SL) * n + i];
    }
    if (p < P) s0 += ws[(size_t)p * n + i];
  }
  red[sl][e] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RED_SL; ++k) t += red[k][e];
    // two destinations (edet_reduce_partials2): columns [0, n_a) -> dst (may be NULL: dropped), [n_a, n) -> dst_b
    float* d = (n_a < 0 || i < n_a) ? (dst ? dst + i : nullptr) : dst_b + (i - n_a);
    if (d) *d = accumulate ? *d + t : t;
  }
}

// grid of a grid-stride elementwise kernel: one thread per item up to one round of what the chip holds of the kernel
inline int ew_grid(int64_t total, const void* fn = nullptr) {
  int64_t g = (total + THREADS - 1) / THREADS;
  int cap = 4096;
  if (fn) {
    const int slots = edet_resident_wgs(fn, THREADS, 0);
    if (slots > 0) cap = slots;
  }
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ---- deferred, batched partial-sum reductions (r04) -----------------------------------------------------------------
// Every weight-gradient kernel of the backward pass ends in "dst += sum of my P partial rows": 192 launches of ~7 us per
// D0 step, each a kernel boundary on the critical chain.  With deferral on for a stream (edet_reduce_defer),
// edet_reduce_partials records (ws, P, n, dst) instead of launching; edet_reduce_flush launches ONE kernel over the
// recorded table (passed by value as kernel arguments, <= RED_BATCH entries per launch: nothing is copied to the device,
// so a flush is legal inside a hipGraph capture).  Entries with the same destination (the tower kernels shared by the
// pyramid levels) form a group that one set of workgroups adds in recording order: a fixed summation order (the same bits on
// every run), though not the immediate kernel's.  The caller keeps the partial regions intact until the flush (edet_reduce_deferred_end tells it how
// far they reach).
namespace {

constexpr int RED_BATCH = 96;
struct RedItem { const float* ws; float* dst; int P; int n; };
struct RedBatch {
  RedItem it[RED_BATCH];
  unsigned short gfirst[RED_BATCH + 1];     // group g = items gfirst[g] .. gfirst[g+1]-1 (same dst, same n)
  int gblock[RED_BATCH + 1];                // prefix sum of the groups' 64-element blocks
  int ngroups;
};

// Workgroup = 64 consecutive elements x 4 row slices: every row is read in whole 256-byte segments (the immediate kernel
// reads 64-byte ones), thread (e, sl) adds the rows sl, sl + 4, ... with four independent running sums, the four slices
// are combined in slice order.  A fixed order -- the same bits on every run -- but not the immediate kernel's order.
constexpr int RB_E = 64, RB_S = 4;
__global__ __launch_bounds__(RB_E * RB_S) void k_reduce_batch(const RedBatch b) {
  __shared__ float red[RB_S][RB_E];
  int g = 0;
  while (g + 1 < b.ngroups && (int)blockIdx.x >= b.gblock[g + 1]) ++g;
  const int e = threadIdx.x & (RB_E - 1), sl = threadIdx.x / RB_E;
  const int64_t n = b.it[b.gfirst[g]].n;
  const int64_t i = (int64_t)(blockIdx.x - b.gblock[g]) * RB_E + e;
  const bool ok = i < n;
  float acc = 0.f;
  float* dst = b.it[b.gfirst[g]].dst;
  if (sl == 0 && ok) acc = dst[i];
  for (int m = b.gfirst[g]; m < b.gfirst[g + 1]; ++m) {
    const float* ws = b.it[m].ws + (ok ? i : 0);
    const int P = b.it[m].P;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = sl;
    for (; p + 3 * RB_S < P; p += 4 * RB_S) {
      s0 += ws[(size_t)p * n];
      s1 += ws[(size_t)(p + RB_S) * n];
      s2 += ws[(size_t)(p + 2 * RB_S) * n];
      s3 += ws[(size_t)(p + 3 * RB_S) * n];
    }
    for (; p < P; p += RB_S) s0 += ws[(size_t)p * n];
    __syncthreads();
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0) acc += (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
  if (sl == 0 && ok) dst[i] = acc;
}

struct DeferState {
  bool on = false;
  std::vector<RedItem> items;
  const unsigned char* hi = nullptr;
};
std::map<hipStream_t, DeferState>& defer_map() {
  static std::map<hipStream_t, DeferState> m;
  return m;
}

int flush_stream(hipStream_t st, DeferState& d) {
  std::vector<RedItem> items;
  items.swap(d.items);
  d.hi = nullptr;
  // groups: entries with the same destination, members in recording order, groups in order of first appearance
  std::vector<std::vector<RedItem>> groups;
  for (const RedItem& r : items) {
    size_t g = 0;
    while (g < groups.size() && groups[g][0].dst != r.dst) ++g;
    if (g == groups.size()) groups.emplace_back();
    groups[g].push_back(r);
  }
  RedBatch b;
  int ni = 0, ng = 0, nblocks = 0;
  auto launch = [&]() -> int {
    if (ni == 0) return 0;
    b.gfirst[ng] = (unsigned short)ni;
    b.gblock[ng] = nblocks;
    b.ngroups = ng;
    edet_launch(k_reduce_batch, dim3((unsigned)nblocks), dim3(RB_E * RB_S), 0, st, b);
    EDET_LAUNCH_CHECK("edet_reduce_flush");
    ni = ng = nblocks = 0;
    return 0;
  };
  memset(&b, 0, sizeof(b));
  for (const std::vector<RedItem>& grp : groups) {
    size_t pos = 0;
    while (pos < grp.size()) {
      const size_t left = grp.size() - pos;
      size_t take = left < (size_t)(RED_BATCH - ni) ? left : (size_t)(RED_BATCH - ni);
      // a group is split over two launches (stream order keeps the sum order) only when it alone exceeds a batch
      if (take == 0 || (take < left && ni > 0)) {
        if (int rc = launch()) return rc;
        continue;
      }
      b.gfirst[ng] = (unsigned short)ni;
      b.gblock[ng] = nblocks;
      for (size_t k = 0; k < take; ++k) b.it[ni++] = grp[pos + k];
      nblocks += (grp[0].n + RB_E - 1) / RB_E;
      ++ng;
      pos += take;
    }
  }
  return launch();
}

}  // namespace

extern "C" int edet_reduce_defer(void* stream, int enable) {
  DeferState& d = defer_map()[to_stream(stream)];
  if (!enable && d.on && !d.items.empty()) {
    if (int rc = flush_stream(to_stream(stream), d)) return rc;
  }
  d.on = enable != 0;
  return 0;
}
extern "C" int edet_reduce_flush(void* stream) {
  auto it = defer_map().find(to_stream(stream));
  if (it == defer_map().end() || it->second.items.empty()) return 0;
  return flush_stream(to_stream(stream), it->second);
}
extern "C" int edet_reduce_deferred_end(void* stream, const void** hi_out) {
  EDET_CHECK(hi_out, "edet_reduce_deferred_end: null pointer");
  auto it = defer_map().find(to_stream(stream));
  *hi_out = it == defer_map().end() ? nullptr : it->second.hi;
  return 0;
}

// dst_a[i] += column i (i < n_a; dst_a may be NULL), dst_b[i - n_a] += column i (n_a <= i < n)
int edet_reduce_partials2(const float* ws, int P, int64_t n, float* dst_a, int64_t n_a, float* dst_b, hipStream_t st) {
  edet_launch(k_reduce_partials, dim3((unsigned)((n + 15) / 16)), dim3(16 * RED_SL), 0, st, ws, P, n, dst_a, 1, n_a, dst_b);
  EDET_LAUNCH_CHECK("edet_reduce_partials2");
  return 0;
}
int edet_reduce_partials(const float* ws, int P, int64_t n, float* dst, hipStream_t st) {
  {
    auto it = defer_map().find(st);
    if (it != defer_map().end() && it->second.on && n < (int64_t)1 << 31) {
      DeferState& d = it->second;
      RedItem r;
      r.ws = ws; r.dst = dst; r.P = P; r.n = (int)n;
      d.items.push_back(r);
      const unsigned char* end = reinterpret_cast<const unsigned char*>(ws) + (size_t)P * n * sizeof(float);
      if (!d.hi || end > d.hi) d.hi = end;
      return 0;
    }
  }
  edet_launch(k_reduce_partials, dim3((unsigned)((n + 15) / 16)), dim3(16 * RED_SL), 0, st, ws, P, n, dst, 1, (int64_t)-1, (float*)nullptr);
  EDET_LAUNCH_CHECK("edet_reduce_partials");
  return 0;
}
// dst[i] = sum of the partial rows (no accumulation)
int edet_reduce_partials_set(const float* ws, int P, int64_t n, float* dst, hipStream_t st) {
  edet_launch(k_reduce_partials, dim3((unsigned)((n + 15) / 16)), dim3(16 * RED_SL), 0, st, ws, P, n, dst, 0, (int64_t)-1, (float*)nullptr);
  EDET_LAUNCH_CHECK("edet_reduce_partials");
  return 0;
}

extern "C" int edet_bn_finalize(const float* partials, int nparts, int c, double count,
                                const float* gamma, const float* beta, float eps, float momentum,
                                int bessel, float* moving_mean, float* moving_var, float* scale, float* shift,
                                float* mean, float* rstd, void* stream) {
  EDET_CHECK(partials && gamma && beta && scale && shift && mean && rstd, "edet_bn_finalize: null pointer");
  edet_launch(k_bn_finalize, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, to_stream(stream), partials, nparts, c, count, gamma, beta, eps,
                                                            momentum, bessel, moving_mean, moving_var, scale, shift,
                                                            mean, rstd);
  EDET_LAUNCH_CHECK("edet_bn_finalize");
  return 0;
}

extern "C" int edet_bn_eval(int c, const float* gamma, const float* beta, float eps,
                            const float* moving_mean, const float* moving_var, float* scale, float* shift,
                            void* stream) {
  EDET_CHECK(gamma && beta && moving_mean && moving_var && scale && shift, "edet_bn_eval: null pointer");
  edet_launch(k_bn_eval, dim3(cdiv(c, 128)), dim3(128), 0, to_stream(stream), c, gamma, beta, eps, moving_mean, moving_var, scale, shift);
  EDET_LAUNCH_CHECK("edet_bn_eval");
  return 0;
}

extern "C" int edet_bn_bwd_reduce(const void* dz, const void* y, int64_t rows, int c, int ld,
                                  const float* mean, const float* rstd, float* stat_partials,
                                  int* nparts_out, int dtype, void* stream) {
  EDET_CHECK(dz && y && mean && rstd && stat_partials, "edet_bn_bwd_reduce: null pointer");
  EDET_CHECK(c % 8 == 0 && ld % 8 == 0 && c <= 6144, "edet_bn_bwd_reduce: c/ld must be multiples of 8, c <= 6144");
  const RowMap m = row_map(c);
  const int grid = persistent_grid(rows, m.rpp, 512);
  if (nparts_out) *nparts_out = grid;
  const size_t lds = (size_t)(2 * c + 2 * THREADS * 8) * sizeof(float);
  if (dtype == EDET_BF16)
    edet_launch(k_bn_bwd_reduce<bf16_t>, grid, dim3(THREADS), lds, to_stream(stream), (const bf16_t*)dz, (const bf16_t*)y, rows, c, ld, mean, rstd, stat_partials, m);
  else if (dtype == EDET_F32)
    edet_launch(k_bn_bwd_reduce<float>, grid, dim3(THREADS), lds, to_stream(stream), (const float*)dz, (const float*)y, rows, c, ld, mean, rstd, stat_partials, m);
  else EDET_CHECK(false, "edet_bn_bwd_reduce: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_bn_bwd_reduce");
  return 0;
}

extern "C" int edet_bn_bwd_finalize(const float* partials, int nparts, int c, double count,
                                    const float* gamma, const float* mean, const float* rstd,
                                    float* dgamma, float* dbeta, float* dbias,
                                    float* a, float* b, float* cc, void* stream) {
  EDET_CHECK(partials && gamma && mean && rstd && a && b && cc, "edet_bn_bwd_finalize: null pointer");
  (void)dbias;  // d(bias before BatchNorm) is analytically zero: BN removes the mean
  edet_launch(k_bn_bwd_finalize, dim3(cdiv(c, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, to_stream(stream), partials, nparts, c, count, gamma, mean, rstd,
                                                                dgamma, dbeta, a, b, cc);
  EDET_LAUNCH_CHECK("edet_bn_bwd_finalize");
  return 0;
}

extern "C" int edet_bn_res(const edet_tview_t* y, const void* residual, void* out, int ldo,
                           int dtype, void* stream) {
  EDET_CHECK(y && y->data && out, "edet_bn_res: null pointer");
  EDET_CHECK(y->c % 8 == 0 && y->ld % 8 == 0 && ldo % 8 == 0, "edet_bn_res: c/ld % 8");
  const int64_t rows = (int64_t)y->n * y->h * y->w;
  const int grid = ew_grid(rows * (y->c / 8), dtype == EDET_BF16 ? reinterpret_cast<const void*>(&k_bn_res<bf16_t>) : nullptr);
  if (dtype == EDET_BF16) edet_launch(k_bn_res<bf16_t>, grid, dim3(THREADS), 0, to_stream(stream), *y, (const bf16_t*)residual, (bf16_t*)out, ldo, rows);
  else if (dtype == EDET_F32) edet_launch(k_bn_res<float>, grid, dim3(THREADS), 0, to_stream(stream), *y, (const float*)residual, (float*)out, ldo, rows);
  else EDET_CHECK(false, "edet_bn_res: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_bn_res");
  return 0;
}

extern "C" int edet_add(void* dst, const void* src, int64_t rows, int c, int ld, int beta,
                        int dtype, void* stream) {
  EDET_CHECK(dst && src, "edet_add: null pointer");
  EDET_CHECK(c % 8 == 0 && ld % 8 == 0, "edet_add: c/ld % 8");
  const int grid = ew_grid(rows * (c / 8), dtype == EDET_BF16 ? reinterpret_cast<const void*>(&k_add<bf16_t>) : nullptr);
  if (dtype == EDET_BF16) edet_launch(k_add<bf16_t>, grid, dim3(THREADS), 0, to_stream(stream), (bf16_t*)dst, (const bf16_t*)src, rows, c, ld, beta);
  else if (dtype == EDET_F32) edet_launch(k_add<float>, grid, dim3(THREADS), 0, to_stream(stream), (float*)dst, (const float*)src, rows, c, ld, beta);
  else EDET_CHECK(false, "edet_add: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_add");
  return 0;
}

static int se_wg_per_img(int n, int hw, int rpp) {
  int passes = cdiv(hw, rpp);
  int w = 1024 / (n > 0 ? n : 1);
  if (w < 1) w = 1;
  if (w > passes) w = passes;
  if (w > 64) w = 64;
  return w;
}

// Rows of one image per k_se_pool workgroup: ~64 K elements (128 KB of bf16), at least four passes of the row map; a
// function of the map size and channel count ONLY, so an image is summed the same way in every batch.  If the chunk rows
// of the whole batch do not fit the caller's scratch the chunks grow (then the order depends on the scratch size; the
// engine's scratch -- EDET_MAX_PARTS * 2 * widest layer floats -- holds every EfficientDet / EfficientNetV2 layer up to
// batch x chunks = 2048 at the widest layer).
static int se_chunk_rows(int hw, int c, int rpp, int n, size_t scratch_floats, int* nchunks_out) {
  int cr = cdiv(cdiv(65536, c), rpp) * rpp;
  if (cr < 4 * rpp) cr = 4 * rpp;
  // at most 64 chunks per image: the consumer adds an image's chunk sums one after the other (r03c: the 768x768 and
  // 384x384 maps of efficientdet-d7x had 400-650 chunks and the FC kernel spent 100 us adding them)
  const int cap = cdiv(cdiv(hw, 64), rpp) * rpp;
  if (cr < cap) cr = cap;
  while ((size_t)n * cdiv(hw, cr) * c > scratch_floats && cr < hw) cr *= 2;
  *nchunks_out = cdiv(hw, cr);
  return cr;
}

static int se_pool_launch(const edet_tview_t* in, float* parts, size_t scratch_floats, int dtype, void* stream,
                          int* nchunks_out) {
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && in->c <= 8192, "edet_se_pool: c/ld");
  const RowMap m = row_map(in->c);
  int nchunks = 1;
  const int cr = se_chunk_rows(in->h * in->w, in->c, m.rpp, in->n, scratch_floats, &nchunks);
  EDET_CHECK((size_t)in->n * nchunks * in->c <= scratch_floats, "edet_se_pool: scratch of %zu floats < %d x %d", scratch_floats,
             in->n, in->c);
  const int cblk = in->c < m.tpr * 8 ? in->c : m.tpr * 8;
  const size_t lds = (size_t)m.rpp * cblk * sizeof(float);
  if (dtype == EDET_BF16) edet_launch(k_se_pool<bf16_t>, dim3(in->n * nchunks), dim3(THREADS), lds, to_stream(stream), *in, parts, nchunks, cr, m);
  else if (dtype == EDET_F32) edet_launch(k_se_pool<float>, dim3(in->n * nchunks), dim3(THREADS), lds, to_stream(stream), *in, parts, nchunks, cr, m);
  else EDET_CHECK(false, "edet_se_pool: bad dtype %d", dtype);
  *nchunks_out = nchunks;
  return 0;
}

extern "C" int edet_se_pool(const edet_tview_t* in, float* pooled_sum, void* scratch, size_t scratch_bytes,
                            int dtype, void* stream) {
  EDET_CHECK(in && in->data && pooled_sum && scratch, "edet_se_pool: null pointer");
  int nchunks = 1;
  if (se_pool_launch(in, (float*)scratch, scratch_bytes / sizeof(float), dtype, stream, &nchunks)) return -1;
  const int total = in->n * in->c;
  edet_launch(k_se_pool_finish, dim3(cdiv(total, THREADS)), dim3(THREADS), 0, to_stream(stream), (const float*)scratch, nchunks, in->c, pooled_sum, total);
  EDET_LAUNCH_CHECK("edet_se_pool");
  return 0;
}

static size_t se_fc_lds(int c, int se, int threads) { return (size_t)(c + se + threads) * sizeof(float); }

extern "C" int edet_se_fc(const float* pooled_sum, int n, int c, int se, float inv_hw,
                          const float* w1, const float* b1, const float* w2, const float* b2,
                          float* hidden_pre, float* gate, int act, void* stream) {
  EDET_CHECK(pooled_sum && w1 && b1 && w2 && b2 && hidden_pre && gate, "edet_se_fc: null pointer");
  EDET_CHECK(act >= EDET_ACT_NONE && act <= EDET_ACT_LAST, "edet_se_fc: activation %d", act);
  const int fc_threads = c >= 512 ? SE_FC_THREADS : THREADS;
  edet_launch(k_se_fc, dim3(n), dim3(fc_threads), se_fc_lds(c, se, fc_threads), to_stream(stream), const_cast<float*>(pooled_sum), (const float*)nullptr, 0, c, se, inv_hw, w1, b1, w2, b2, hidden_pre, gate, act);
  EDET_LAUNCH_CHECK("edet_se_fc");
  return 0;
}

extern "C" int edet_se_squeeze_excite(const edet_tview_t* in, void* scratch, size_t scratch_bytes, int se,
                                      float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
                                      float* pooled_sum, float* hidden_pre, float* gate, int act, int dtype,
                                      void* stream) {
  EDET_CHECK(in && in->data && scratch && pooled_sum && w1 && b1 && w2 && b2 && hidden_pre && gate,
             "edet_se_squeeze_excite: null pointer");
  EDET_CHECK(act >= EDET_ACT_NONE && act <= EDET_ACT_LAST, "edet_se_squeeze_excite: activation %d", act);
  int nchunks = 1;
  if (se_pool_launch(in, (float*)scratch, scratch_bytes / sizeof(float), dtype, stream, &nchunks)) return -1;
  const int c = in->c;
  // wide blocks: the sliced pair of kernels, if the partial hidden sums fit behind the pooling's chunk sums
  const int nslice = cdiv(c, SE_SLICE);
  const size_t used = ((size_t)in->n * nchunks * c + 63) / 64 * 64;
  if ((int64_t)c * se >= SE_SPLIT_MIN && used + (size_t)in->n * nslice * se <= scratch_bytes / sizeof(float)) {
    float* hpart = (float*)scratch + used;
    const int nblk = cdiv(in->n, SE_IB);
    edet_launch(k_se_fc1_split, dim3(nslice, nblk), dim3(THREADS), 0, to_stream(stream), pooled_sum, (const float*)scratch, nchunks, c, se, inv_hw, w1, hpart, nslice, in->n);
    edet_launch(k_se_fc2_split, dim3(nslice, nblk), dim3(THREADS), (size_t)SE_IB * se * sizeof(float), to_stream(stream), (const float*)hpart, nslice, c, se, b1, w2, b2, hidden_pre, gate, act, in->n);
    EDET_LAUNCH_CHECK("edet_se_squeeze_excite");
    return 0;
  }
  const int fc_threads = c >= 512 ? SE_FC_THREADS : THREADS;
  edet_launch(k_se_fc, dim3(in->n), dim3(fc_threads), se_fc_lds(c, se, fc_threads), to_stream(stream), pooled_sum, (const float*)scratch, nchunks, c, se, inv_hw, w1, b1, w2, b2, hidden_pre, gate, act);
  EDET_LAUNCH_CHECK("edet_se_squeeze_excite");
  return 0;
}

extern "C" int edet_se_fc_bwd(const float* pooled_sum, const float* hidden_pre, const float* gate,
                              const float* dgate, int n, int c, int se, float inv_hw,
                              const float* w1, const float* w2,
                              float* dw1, float* db1, float* dw2, float* db2,
                              float* dpool, float* scratch, int act, void* stream) {
  EDET_CHECK(act >= EDET_ACT_NONE && act <= EDET_ACT_LAST, "edet_se_fc_bwd: activation %d", act);
  EDET_CHECK(pooled_sum && hidden_pre && gate && dgate && w1 && w2 && dw1 && db1 && dw2 && db2 && dpool && scratch,
             "edet_se_fc_bwd: null pointer");
  if ((int64_t)c * se >= SE_SPLIT_MIN_BWD) {      // wide blocks: sliced over the channel axis (the scratch holds [n][slices][se] more)
    const int nslice = cdiv(c, SE_SLICE);
    float* hpart = scratch + (size_t)n * (c + 2 * se);
    edet_launch(k_se_fc_bwd_img1, dim3(nslice, n), dim3(THREADS), 0, to_stream(stream), gate, dgate, n, c, se, w2, scratch, hpart, nslice);
    edet_launch(k_se_fc_bwd_img2, dim3(nslice, n), dim3(THREADS), (size_t)se * sizeof(float), to_stream(stream), hidden_pre, (const float*)hpart, nslice, n, c, se, inv_hw, w1, dpool, scratch, act);
  } else {
    edet_launch(k_se_fc_bwd_img, dim3(n), dim3(c >= 512 ? SE_FC_THREADS : THREADS), (size_t)(c + se) * sizeof(float), to_stream(stream), hidden_pre, gate, dgate, n, c, se, inv_hw, w1, w2, dpool, scratch, act);
  }
  const int nsplit = n >= 2 * SE_SPLIT ? SE_SPLIT : 1;
  const int per_split = cdiv(n, nsplit);
  const int nsl = cdiv(n, per_split);
  // the image slices' partials live behind the per-image vectors of the scratch (see the header for its size)
  float* parts = scratch + (size_t)n * (c + (2 + cdiv(c, 128)) * se);
  edet_launch(k_se_fc_bwd_par, dim3(cdiv(c, 64), nsl, cdiv(se, SE_JB)), dim3(THREADS), (size_t)2 * SE_NB * SE_JB * sizeof(float),
              to_stream(stream), pooled_sum, (const float*)scratch, n, c, se, inv_hw, parts, per_split);
  const size_t total = (size_t)2 * c * se + se + c;
  edet_launch(k_se_fc_bwd_sum, dim3(cdiv(total, THREADS)), dim3(THREADS), 0, to_stream(stream), (const float*)parts, nsl, c, se,
              dw1, db1, dw2, db2);
  EDET_LAUNCH_CHECK("edet_se_fc_bwd");
  return 0;
}

extern "C" int edet_se_gate_bwd(const edet_tview_t* in, void* g, const float* dpool,
                                const float* mean, const float* rstd,
                                float* stat_partials, int* nparts_out, int dtype, void* stream) {
  EDET_CHECK(in && in->data && in->gate && g && dpool && mean && rstd && stat_partials, "edet_se_gate_bwd: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && in->c <= 6144, "edet_se_gate_bwd: c/ld (c <= 6144)");
  const RowMap m = row_map(in->c);
  int wpi = se_wg_per_img(in->n, in->h * in->w, m.rpp);
  while (in->n * wpi > EDET_MAX_PARTS && wpi > 1) --wpi;
  EDET_CHECK(in->n * wpi <= EDET_MAX_PARTS, "edet_se_gate_bwd: batch %d exceeds %d partial rows", in->n, EDET_MAX_PARTS);
  if (nparts_out) *nparts_out = in->n * wpi;
  const size_t lds = (size_t)(2 * in->c + 2 * THREADS * 8) * sizeof(float);
  const bool other = in->act > EDET_ACT_SWISH;
  if (dtype == EDET_BF16 && !other) edet_launch(k_se_gate_bwd<bf16_t, false>, dim3(in->n * wpi), dim3(THREADS), lds, to_stream(stream), *in, (bf16_t*)g, dpool, mean, rstd, stat_partials, wpi, m);
  else if (dtype == EDET_BF16) edet_launch(k_se_gate_bwd<bf16_t, true>, dim3(in->n * wpi), dim3(THREADS), lds, to_stream(stream), *in, (bf16_t*)g, dpool, mean, rstd, stat_partials, wpi, m);
  else if (dtype == EDET_F32 && !other) edet_launch(k_se_gate_bwd<float, false>, dim3(in->n * wpi), dim3(THREADS), lds, to_stream(stream), *in, (float*)g, dpool, mean, rstd, stat_partials, wpi, m);
  else if (dtype == EDET_F32) edet_launch(k_se_gate_bwd<float, true>, dim3(in->n * wpi), dim3(THREADS), lds, to_stream(stream), *in, (float*)g, dpool, mean, rstd, stat_partials, wpi, m);
  else EDET_CHECK(false, "edet_se_gate_bwd: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_se_gate_bwd");
  return 0;
}
