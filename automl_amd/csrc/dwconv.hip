// Depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF 'SAME' asymmetric padding), NHWC.
//
// HBM-bound stencil (about 5 FLOP/B): the input tile (with halo) is read once with 16-byte
// coalesced loads, BatchNorm + swish of the *producing* layer are applied on load, the activated
// tile is kept in LDS as fp32 and every thread computes a 1x4 strip of output pixels for 4
// channels from it (sliding-window reuse in registers).  BatchNorm statistic partials of the
// output are produced by the same kernel (one row per persistent workgroup slot).
//
// Reference call sites: efficientdet/backbone/efficientnet_model.py:320-327 (MBConv depthwise),
// efficientdet/tf2/efficientdet_keras.py:195-207,459-464,546-556 (depthwise half of SeparableConv2D).
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int STRIP = 4;  // output pixels per thread along W
constexpr int MAXP = 256; // persistent workgroup slots per channel chunk (<= EDET_MAX_PARTS)

struct DwArgs {
  edet_tview_t in;     // activated input view (fwd, wgrad) / chain target (dgrad)
  edet_gview_t gy;     // dy (dgrad, wgrad)
  const float* w;      // [K][K][C] fp32
  void* out; int ldo;  // fwd output
  int oh, ow;          // conv output geometry
  int pad_t, pad_l;
  int th, tw, cc;      // tile: th x tw pixels (of the kernel's own output space), cc channels
  int tiles_y, tiles_x, nchunks, nsp, P;
  float* stat_partials;
  edet_bwd_epi_t epi;  // dgrad
  float* dweight;      // wgrad
  float* ws;           // wgrad with P > 1 workgroup slots: partial rows [P][K*K][C]
};

__device__ __forceinline__ void decode_tile(const DwArgs& a, int sp, int& n, int& ty0, int& tx0) {
  const int per_img = a.tiles_y * a.tiles_x;
  n = sp / per_img;
  const int r = sp - n * per_img;
  ty0 = (r / a.tiles_x) * a.th;
  tx0 = (r % a.tiles_x) * a.tw;
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int K, int S>
__global__ __launch_bounds__(THREADS) void k_dw_fwd(const DwArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int IH = (a.th - 1) * S + K, IW = (a.tw - 1) * S + K;
  float* tile = lds;                       // [IH][IW][cc]
  float* red = lds + IH * IW * a.cc;       // [2][cc]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x % a.nchunks, p = blockIdx.x / a.nchunks;
  const int c0 = chunk * a.cc;
  const int C = a.in.c;
  const int nquad = a.cc / 4;
  const int quad = tid % nquad;            // fixed per thread (THREADS % nquad == 0)
  const int cq = c0 + quad * 4;
  const bool q_ok = cq < C;
  const bool want_stats = a.stat_partials != nullptr;

  float wreg[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    if (q_ok) load4<float>(a.w + (size_t)t * C + cq, wreg[t]);
    else wreg[t][0] = wreg[t][1] = wreg[t][2] = wreg[t][3] = 0.f;
  }
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int i = tid; i < 2 * a.cc; i += THREADS) red[i] = 0.f;

  const int nvec = a.cc / 8;
  const int strips = a.tw / STRIP;
  const int ntask = nquad * strips * a.th;

  for (int sp = p; sp < a.nsp; sp += a.P) {
    int n, oy0, ox0;
    decode_tile(a, sp, n, oy0, ox0);
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
    __syncthreads();
    // ---- stage the activated input tile
    for (int q = tid; q < IH * IW * nvec; q += THREADS) {
      const int v = q % nvec, pix = q / nvec;
      const int ly = pix / IW, lx = pix - ly * IW;
      const int gy = iy0 + ly, gx = ix0 + lx;
      const int c = c0 + v * 8;
      float x[8];
      if (gy >= 0 && gy < a.in.h && gx >= 0 && gx < a.in.w && c < C) {
        load8<T>(reinterpret_cast<const T*>(a.in.data) + ((size_t)(n * a.in.h + gy) * a.in.w + gx) * a.in.ld + c, x);
        ViewCoef vc;
        view_load_coef(a.in, c, vc);
        view_apply(a.in, vc, c, n, x);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      store8<float>(&tile[(size_t)pix * a.cc + v * 8], x);
    }
    __syncthreads();
    // ---- stencil
    for (int q = tid; q < ntask; q += THREADS) {
      const int rest = q / nquad;
      const int sx = rest % strips, ty = rest / strips;
      const int oy = oy0 + ty, ox = ox0 + sx * STRIP;
      if (oy >= a.oh || ox >= a.ow || !q_ok) continue;
      float acc[STRIP][4];
#pragma unroll
      for (int pp = 0; pp < STRIP; ++pp) acc[pp][0] = acc[pp][1] = acc[pp][2] = acc[pp][3] = 0.f;
      constexpr int NCOL = (STRIP - 1) * S + K;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float* rowp = &tile[((size_t)(ty * S + ky) * IW + sx * STRIP * S) * a.cc + quad * 4];
        float col[NCOL][4];
#pragma unroll
        for (int ci = 0; ci < NCOL; ++ci) {
          const float4 t4 = *reinterpret_cast<const float4*>(rowp + (size_t)ci * a.cc);
          col[ci][0] = t4.x; col[ci][1] = t4.y; col[ci][2] = t4.z; col[ci][3] = t4.w;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int pp = 0; pp < STRIP; ++pp)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[pp][e] = fmaf(col[pp * S + kx][e], wreg[ky * K + kx][e], acc[pp][e]);
      }
#pragma unroll
      for (int pp = 0; pp < STRIP; ++pp) {
        if (ox + pp < a.ow) {
          store4<T>(reinterpret_cast<T*>(a.out) + ((size_t)(n * a.oh + oy) * a.ow + ox + pp) * a.ldo + cq, acc[pp]);
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += acc[pp][e]; s2[e] += acc[pp][e] * acc[pp][e]; }
          }
        }
      }
    }
  }
  if (want_stats) {
    // the THREADS / nquad threads of a channel quad: xor butterfly inside the wave, then the waves one after the other
    // (a fixed order, no LDS atomics: the same partial row on every run)
    wave_group_sum(s1, nquad);
    wave_group_sum(s2, nquad);
    __syncthreads();
    for (int wv = 0; wv < THREADS / 64; ++wv) {
      if ((tid >> 6) == wv && (tid & 63) < nquad && q_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[quad * 4 + e] += s1[e];
          red[a.cc + quad * 4 + e] += s2[e];
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < 2 * a.cc; i += THREADS) {
      const int which = i / a.cc, cl = i - which * a.cc;
      if (c0 + cl < C) a.stat_partials[((size_t)p * 2 + which) * C + c0 + cl] = red[i];
    }
  }
}

// ------------------------------------------------------------------------------ data gradient
// Tiles are over the *input* pixels; LDS holds the dy tile those pixels can touch.
template <typename T, int K, int S>
__global__ __launch_bounds__(THREADS) void k_dw_bwd_data(const DwArgs a) {
  extern __shared__ __align__(16) float lds[];
  // dy rows needed by input rows [iy0, iy0+th): oy in [floor((iy0+pad-(K-1))/S) .. (iy0+th-1+pad)/S]
  const int DH = (a.th + K - 2) / S + 2, DW = (a.tw + K - 2) / S + 2;
  float* tile = lds;                       // [DH][DW][cc]
  float* red = lds + DH * DW * a.cc;       // [2][cc]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x % a.nchunks, p = blockIdx.x / a.nchunks;
  const int c0 = chunk * a.cc;
  const int C = a.in.c;
  const int nquad = a.cc / 4;
  const int quad = tid % nquad;
  const int cq = c0 + quad * 4;
  const bool q_ok = cq < C;
  const bool want_stats = a.epi.stat_partials != nullptr;

  float wreg[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    if (q_ok) load4<float>(a.w + (size_t)t * C + cq, wreg[t]);
    else wreg[t][0] = wreg[t][1] = wreg[t][2] = wreg[t][3] = 0.f;
  }
  float sc[4] = {1, 1, 1, 1}, sh[4] = {0, 0, 0, 0}, mean[4] = {0, 0, 0, 0}, rstd[4] = {1, 1, 1, 1};
  if (q_ok) {
    if (a.in.scale) { load4<float>(a.in.scale + cq, sc); load4<float>(a.in.shift + cq, sh); }
    if (want_stats) { load4<float>(a.epi.mean + cq, mean); load4<float>(a.epi.rstd + cq, rstd); }
  }
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int i = tid; i < 2 * a.cc; i += THREADS) red[i] = 0.f;

  const int nvec = a.cc / 8;
  const int strips = a.tw / STRIP;
  const int ntask = nquad * strips * a.th;

  for (int sp = p; sp < a.nsp; sp += a.P) {
    int n, iy0, ix0;
    decode_tile(a, sp, n, iy0, ix0);
    // first dy row/col that can contribute (floor division for possibly negative numerators)
    const int ny = iy0 + a.pad_t - (K - 1), nx = ix0 + a.pad_l - (K - 1);
    const int dy0 = ny >= 0 ? ny / S : -((-ny + S - 1) / S);
    const int dx0 = nx >= 0 ? nx / S : -((-nx + S - 1) / S);
    __syncthreads();
    for (int q = tid; q < DH * DW * nvec; q += THREADS) {
      const int v = q % nvec, pix = q / nvec;
      const int ly = pix / DW, lx = pix - ly * DW;
      const int oy = dy0 + ly, ox = dx0 + lx;
      const int c = c0 + v * 8;
      float g[8];
      if (oy >= 0 && oy < a.oh && ox >= 0 && ox < a.ow && c < C) {
        GradCoef gc;
        grad_load_coef(a.gy, c, gc);
        grad_load<T>(a.gy, gc, ((size_t)(n * a.oh + oy) * a.ow + ox) * a.gy.ld + c, g);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
      }
      store8<float>(&tile[(size_t)pix * a.cc + v * 8], g);
    }
    __syncthreads();
    for (int q = tid; q < ntask; q += THREADS) {
      const int rest = q / nquad;
      const int sx = rest % strips, ty = rest / strips;
      const int iy = iy0 + ty, ixb = ix0 + sx * STRIP;
      if (iy >= a.in.h || ixb >= a.in.w || !q_ok) continue;
      float acc[STRIP][4];
#pragma unroll
      for (int pp = 0; pp < STRIP; ++pp) acc[pp][0] = acc[pp][1] = acc[pp][2] = acc[pp][3] = 0.f;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int t = iy + a.pad_t - ky;
        if (t < 0 || (t % S) != 0) continue;
        const int oy = t / S;
        if (oy >= a.oh) continue;
        const int ly = oy - dy0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int pp = 0; pp < STRIP; ++pp) {
            const int u = ixb + pp + a.pad_l - kx;
            if (u < 0 || (u % S) != 0) continue;
            const int ox = u / S;
            if (ox >= a.ow) continue;
            const float4 t4 = *reinterpret_cast<const float4*>(
                &tile[((size_t)ly * DW + (ox - dx0)) * a.cc + quad * 4]);
            acc[pp][0] = fmaf(t4.x, wreg[ky * K + kx][0], acc[pp][0]);
            acc[pp][1] = fmaf(t4.y, wreg[ky * K + kx][1], acc[pp][1]);
            acc[pp][2] = fmaf(t4.z, wreg[ky * K + kx][2], acc[pp][2]);
            acc[pp][3] = fmaf(t4.w, wreg[ky * K + kx][3], acc[pp][3]);
          }
        }
      }
      // epilogue: chain through the input view
#pragma unroll
      for (int pp = 0; pp < STRIP; ++pp) {
        const int ix = ixb + pp;
        if (ix >= a.in.w) continue;
        const size_t off = ((size_t)(n * a.in.h + iy) * a.in.w + ix) * a.in.ld + cq;
        float g[4] = {acc[pp][0], acc[pp][1], acc[pp][2], acc[pp][3]};
        float x[4] = {0, 0, 0, 0};
        if (a.in.act != EDET_ACT_NONE || want_stats) load4<T>(reinterpret_cast<const T*>(a.in.data) + off, x);
        if (a.in.act != EDET_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] *= act_grad_(a.in.act, fmaf(x[e], sc[e], sh[e]));
        }
        if (a.epi.beta) {
          float old[4];
          load4<T>(reinterpret_cast<const T*>(a.epi.gout) + off, old);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        store4<T>(reinterpret_cast<T*>(a.epi.gout) + off, g);
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += g[e]; s2[e] += g[e] * (x[e] - mean[e]) * rstd[e]; }
        }
      }
    }
  }
  if (want_stats) {
    // the THREADS / nquad threads of a channel quad: xor butterfly inside the wave, then the waves one after the other
    // (a fixed order, no LDS atomics: the same partial row on every run)
    wave_group_sum(s1, nquad);
    wave_group_sum(s2, nquad);
    __syncthreads();
    for (int wv = 0; wv < THREADS / 64; ++wv) {
      if ((tid >> 6) == wv && (tid & 63) < nquad && q_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[quad * 4 + e] += s1[e];
          red[a.cc + quad * 4 + e] += s2[e];
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < 2 * a.cc; i += THREADS) {
      const int which = i / a.cc, cl = i - which * a.cc;
      if (c0 + cl < C) a.epi.stat_partials[((size_t)p * 2 + which) * C + c0 + cl] = red[i];
    }
  }
}

// ---------------------------------------------------------------------------- weight gradient
template <typename T, int K, int S>
__global__ __launch_bounds__(THREADS) void k_dw_bwd_weight(const DwArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int IH = (a.th - 1) * S + K, IW = (a.tw - 1) * S + K;
  float* tile = lds;                       // [IH][IW][cc]
  float* red = lds + IH * IW * a.cc;       // [K*K][cc]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x % a.nchunks, p = blockIdx.x / a.nchunks;
  const int c0 = chunk * a.cc;
  const int C = a.in.c;
  const int nquad = a.cc / 4;
  const int quad = tid % nquad;
  const int cq = c0 + quad * 4;
  const bool q_ok = cq < C;

  float wacc[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wacc[t][0] = wacc[t][1] = wacc[t][2] = wacc[t][3] = 0.f;
  for (int i = tid; i < K * K * a.cc; i += THREADS) red[i] = 0.f;
  float ga[4] = {1, 1, 1, 1}, gb[4] = {0, 0, 0, 0}, gcc[4] = {0, 0, 0, 0};
  if (q_ok && a.gy.a) {
    load4<float>(a.gy.a + cq, ga);
    load4<float>(a.gy.b + cq, gb);
    load4<float>(a.gy.cc + cq, gcc);
  }

  const int nvec = a.cc / 8;
  const int strips = a.tw / STRIP;
  const int ntask = nquad * strips * a.th;

  for (int sp = p; sp < a.nsp; sp += a.P) {
    int n, oy0, ox0;
    decode_tile(a, sp, n, oy0, ox0);
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
    __syncthreads();
    for (int q = tid; q < IH * IW * nvec; q += THREADS) {
      const int v = q % nvec, pix = q / nvec;
      const int ly = pix / IW, lx = pix - ly * IW;
      const int gy = iy0 + ly, gx = ix0 + lx;
      const int c = c0 + v * 8;
      float x[8];
      if (gy >= 0 && gy < a.in.h && gx >= 0 && gx < a.in.w && c < C) {
        load8<T>(reinterpret_cast<const T*>(a.in.data) + ((size_t)(n * a.in.h + gy) * a.in.w + gx) * a.in.ld + c, x);
        ViewCoef vc;
        view_load_coef(a.in, c, vc);
        view_apply(a.in, vc, c, n, x);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      store8<float>(&tile[(size_t)pix * a.cc + v * 8], x);
    }
    __syncthreads();
    for (int q = tid; q < ntask; q += THREADS) {
      const int rest = q / nquad;
      const int sx = rest % strips, ty = rest / strips;
      const int oy = oy0 + ty, ox = ox0 + sx * STRIP;
      if (oy >= a.oh || ox >= a.ow || !q_ok) continue;
      float g[STRIP][4];
#pragma unroll
      for (int pp = 0; pp < STRIP; ++pp) {
        g[pp][0] = g[pp][1] = g[pp][2] = g[pp][3] = 0.f;
        if (ox + pp < a.ow) {
          const size_t off = ((size_t)(n * a.oh + oy) * a.ow + ox + pp) * a.gy.ld + cq;
          load4<T>(reinterpret_cast<const T*>(a.gy.dz) + off, g[pp]);
          if (a.gy.a) {
            float y[4];
            load4<T>(reinterpret_cast<const T*>(a.gy.y) + off, y);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[pp][e] = fmaf(ga[e], g[pp][e], fmaf(gb[e], y[e], gcc[e]));
          }
        }
      }
      constexpr int NCOL = (STRIP - 1) * S + K;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float* rowp = &tile[((size_t)(ty * S + ky) * IW + sx * STRIP * S) * a.cc + quad * 4];
        float col[NCOL][4];
#pragma unroll
        for (int ci = 0; ci < NCOL; ++ci) {
          const float4 t4 = *reinterpret_cast<const float4*>(rowp + (size_t)ci * a.cc);
          col[ci][0] = t4.x; col[ci][1] = t4.y; col[ci][2] = t4.z; col[ci][3] = t4.w;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int pp = 0; pp < STRIP; ++pp)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              wacc[ky * K + kx][e] = fmaf(col[pp * S + kx][e], g[pp][e], wacc[ky * K + kx][e]);
      }
    }
  }
  // the threads of a channel quad: xor butterfly inside the wave, the waves one after the other, then ONE writer per
  // element -- this workgroup slot's partial row in the workspace (edet_reduce_partials adds the rows in order) or, with
  // a single slot, dW itself.  No atomics: the same gradient on every run.
#pragma unroll
  for (int t = 0; t < K * K; ++t) wave_group_sum(wacc[t], nquad);
  __syncthreads();
  for (int wv = 0; wv < THREADS / 64; ++wv) {
    if ((tid >> 6) == wv && (tid & 63) < nquad && q_ok) {
#pragma unroll
      for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[t * a.cc + quad * 4 + e] += wacc[t][e];
    }
    __syncthreads();
  }
  for (int i = tid; i < K * K * a.cc; i += THREADS) {
    const int t = i / a.cc, cl = i - t * a.cc;
    if (c0 + cl < C) {
      if (a.ws) a.ws[((size_t)p * K * K + t) * C + c0 + cl] = red[i];
      else a.dweight[(size_t)t * C + c0 + cl] += red[i];
    }
  }
}

// ------------------------------------------------------------------------------------- host
enum { DW_FWD = 0, DW_BWD_DATA = 1, DW_BWD_WEIGHT = 2 };

// chooses the tile and returns dynamic LDS bytes; space_h/w = extent of the tiled pixel space
size_t plan(DwArgs& a, int which, int k, int s, int n, int space_h, int space_w, int C) {
  a.cc = C >= 32 ? 32 : (C >= 16 ? 16 : 8);
  // channel chunk must split into float4 quads with THREADS % (cc/4) == 0 -> cc in {8,16,32}
  a.tw = space_w >= 16 ? 16 : (space_w + 3) / 4 * 4;
  a.th = 8;
  if (which != DW_BWD_DATA && s == 2) a.th = 4;
  if (space_h < a.th) a.th = space_h;
  a.tiles_y = cdiv(space_h, a.th);
  a.tiles_x = cdiv(space_w, a.tw);
  a.nchunks = cdiv(C, a.cc);
  a.nsp = n * a.tiles_y * a.tiles_x;
  // ~4096 workgroups in flight per launch (256 CUs x several per CU); P <= EDET_MAX_PARTS partial rows
  int P = 4096 / a.nchunks;
  if (P < MAXP) P = MAXP;
  if (P > EDET_MAX_PARTS) P = EDET_MAX_PARTS;
  if (P > a.nsp) P = a.nsp;
  a.P = P;
  size_t elems;
  if (which == DW_BWD_DATA) {
    const int DH = (a.th + k - 2) / s + 2, DW = (a.tw + k - 2) / s + 2;
    elems = (size_t)DH * DW * a.cc + 2 * a.cc;
  } else {
    const int IH = (a.th - 1) * s + k, IW = (a.tw - 1) * s + k;
    elems = (size_t)IH * IW * a.cc + (which == DW_FWD ? 2 : k * k) * a.cc;
  }
  return elems * sizeof(float);
}

template <typename T, int K, int S>
void launch(int which, const DwArgs& a, size_t ldsb, hipStream_t st) {
  const dim3 grid(a.P * a.nchunks), block(THREADS);
  if (which == DW_FWD) edet_launch(k_dw_fwd<T, K, S>, grid, block, ldsb, st, a);
  else if (which == DW_BWD_DATA) edet_launch(k_dw_bwd_data<T, K, S>, grid, block, ldsb, st, a);
  else edet_launch(k_dw_bwd_weight<T, K, S>, grid, block, ldsb, st, a);
}

template <typename T>
int dispatch(int which, int k, int s, const DwArgs& a, size_t ldsb, hipStream_t st) {
  if (k == 3 && s == 1) launch<T, 3, 1>(which, a, ldsb, st);
  else if (k == 3 && s == 2) launch<T, 3, 2>(which, a, ldsb, st);
  else if (k == 5 && s == 1) launch<T, 5, 1>(which, a, ldsb, st);
  else if (k == 5 && s == 2) launch<T, 5, 2>(which, a, ldsb, st);
  else EDET_CHECK(false, "depthwise conv: unsupported kernel %d stride %d", k, s);
  EDET_LAUNCH_CHECK("edet_dw");
  return 0;
}

int run(int which, int k, int s, DwArgs& a, int dtype, void* stream, int* nparts_out, void* workspace = nullptr,
        size_t workspace_bytes = 0) {
  const edet_tview_t& in = a.in;
  EDET_CHECK(in.c % 8 == 0 && in.ld % 8 == 0, "depthwise conv: c (%d) and ld (%d) must be multiples of 8", in.c, in.ld);
  a.oh = same_out(in.h, s);
  a.ow = same_out(in.w, s);
  a.pad_t = same_pad_before(in.h, k, s);
  a.pad_l = same_pad_before(in.w, k, s);
  const bool over_input = which == DW_BWD_DATA;
  const size_t ldsb = plan(a, which, k, s, in.n, over_input ? in.h : a.oh, over_input ? in.w : a.ow, in.c);
  EDET_CHECK(ldsb <= 64 * 1024, "depthwise conv: LDS plan too large (%zu bytes)", ldsb);
  if (which == DW_BWD_WEIGHT) {
    // the workgroup slots hand their partial sums over through the workspace; without one (or with one too small for
    // two rows) a single slot per channel chunk adds into dW directly -- slower, the same fixed summation order
    const size_t row_bytes = (size_t)k * k * in.c * sizeof(float);
    const size_t ws_rows = workspace ? workspace_bytes / row_bytes : 0;
    if ((size_t)a.P > ws_rows) a.P = ws_rows >= 2 ? (int)ws_rows : 1;
    a.ws = a.P > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  }
  if (nparts_out) *nparts_out = a.P;
  int rc;
  if (dtype == EDET_BF16) rc = dispatch<bf16_t>(which, k, s, a, ldsb, to_stream(stream));
  else if (dtype == EDET_F32) rc = dispatch<float>(which, k, s, a, ldsb, to_stream(stream));
  else EDET_CHECK(false, "depthwise conv: bad dtype %d", dtype);
  if (rc == 0 && which == DW_BWD_WEIGHT && a.ws &&
      edet_reduce_partials(a.ws, a.P, (int64_t)k * k * in.c, a.dweight, to_stream(stream)) != 0) return -2;
  return rc;
}

}  // namespace

// bf16 row-marching kernels (dw_march.hip): 1 = handled, 0 = not applicable
int dwm_try_fwd(const edet_tview_t* in, const float* weight, int k, int s, void* out, int ldo,
                float* stat_partials, int* nparts_out, hipStream_t st);
int dwm_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, int k, int s, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st);
int dwm_try_dgrad(const edet_gview_t* dy, const float* weight, int k, int s, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st);

int dwm_try_bwd_fused(const edet_gview_t* dy, const float* weight, int k, int s, const edet_tview_t* in,
                      const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                      size_t workspace_bytes, hipStream_t st);

extern "C" int edet_dw_fwd(const edet_tview_t* in, const float* weight, int k, int stride,
                           void* out, int ldo, float* stat_partials, int* nparts_out,
                           int dtype, void* stream) {
  EDET_CHECK(in && in->data && weight && out, "edet_dw_fwd: null pointer");
  EDET_CHECK(ldo % 4 == 0 && ldo >= in->c, "edet_dw_fwd: bad ldo %d", ldo);
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.w = weight; a.out = out; a.ldo = ldo; a.stat_partials = stat_partials;
  if (dtype == EDET_BF16 && ldo % 8 == 0) {
    const int rc = dwm_try_fwd(in, weight, k, stride, out, ldo, stat_partials, nparts_out, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  return run(DW_FWD, k, stride, a, dtype, stream, nparts_out);
}

extern "C" int edet_dw_bwd_data(const edet_gview_t* dy, const float* weight, int k, int stride,
                                const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                                int dtype, void* stream) {
  EDET_CHECK(dy && dy->dz && weight && in && in->data && epi && epi->gout, "edet_dw_bwd_data: null pointer");
  EDET_CHECK(!(epi->stat_partials && epi->beta), "edet_dw_bwd_data: fused stats need beta == 0");
  EDET_CHECK(!in->gate && !epi->dgate, "edet_dw_bwd_data: gated inputs are not supported");
  EDET_CHECK(dy->ld % 8 == 0, "edet_dw_bwd_data: dy ld % 8");
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.gy = *dy; a.w = weight; a.epi = *epi;
  if (dtype == EDET_BF16) {
    const int rc = dwm_try_dgrad(dy, weight, k, stride, in, epi, nparts_out, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  return run(DW_BWD_DATA, k, stride, a, dtype, stream, nparts_out);
}

extern "C" int edet_dw_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, int k, int stride,
                                  float* dweight, void* workspace, size_t workspace_bytes, int dtype,
                                  void* stream) {
  EDET_CHECK(in && in->data && dy && dy->dz && dweight, "edet_dw_bwd_weight: null pointer");
  EDET_CHECK(dy->ld % 4 == 0, "edet_dw_bwd_weight: dy ld % 4");
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.gy = *dy; a.dweight = dweight;
  if (dtype == EDET_BF16 && dy->ld % 8 == 0) {
    const int rc = dwm_try_wgrad(in, dy, k, stride, dweight, workspace, workspace_bytes, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  return run(DW_BWD_WEIGHT, k, stride, a, dtype, stream, nullptr, workspace, workspace_bytes);
}

// Data gradient and weight gradient of one depthwise layer.  Stride 1, bf16: one fused kernel (dw_march.hip,
// k_bwd_fused) that reads (dz, y, x) once; everything else: the two separate entry points, in this order.
extern "C" int edet_dw_bwd(const edet_gview_t* dy, const float* weight, int k, int stride,
                           const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                           float* dweight, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  EDET_CHECK(dy && dy->dz && weight && in && in->data && epi && epi->gout && dweight, "edet_dw_bwd: null pointer");
  EDET_CHECK(!(epi->stat_partials && epi->beta), "edet_dw_bwd: fused stats need beta == 0");
  if (dtype == EDET_BF16 && dy->ld % 8 == 0) {
    const int rc = dwm_try_bwd_fused(dy, weight, k, stride, in, epi, nparts_out, dweight, workspace,
                                     workspace_bytes, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  const int rc = edet_dw_bwd_weight(in, dy, k, stride, dweight, workspace, workspace_bytes, dtype, stream);
  if (rc != 0) return rc;
  return edet_dw_bwd_data(dy, weight, k, stride, in, epi, nparts_out, dtype, stream);
}
