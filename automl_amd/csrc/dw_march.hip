// Depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF 'SAME') for the bf16 path: row-marching
// kernels with an LDS row exchange.
//
// A depthwise conv moves ~4 bytes per output element and does k*k FMAs on it: pure HBM streaming.  Each
// thread owns one output column and CPT consecutive channels and marches DOWN the image: per input row it
// loads the k horizontally adjacent pixels it needs (the neighbouring threads read the same cache lines, so
// every byte comes from HBM once and k times from L1), applies the producer's BatchNorm + swish on load and
// accumulates into a rotating set of ceil(k/stride) output-row accumulators held in registers; a finished
// output row is written once.  The k*k weights of the thread's channels stay in registers (fp32) for the whole
// kernel; CPT = 4 (k = 3) or 2 (k = 5) keeps the kernel near 100 VGPRs = 4-5 waves per SIMD, which is what
// hides the HBM latency here (there is no barrier anywhere in the main loop).
//
// The same marching structure gives the weight gradient (accumulate in[r][x+kx] * dy[oy][x] into k*k
// register sums per thread, reduced per workgroup into a workspace slab, summed by a second kernel) and
// the data gradient (march over dy rows, scatter into the in-flight input-row accumulators, then chain
// through act'(z), the optional accumulate and the BatchNorm backward sums in the row epilogue).  Round 6: both
// gradients of a layer come from ONE march (k_bwd_one, any stride; edet_dw_bwd), the separate kernels stay behind the
// separate entry points.
//
// Reference call sites: efficientdet/backbone/efficientnet_model.py:320-327 (MBConv DepthwiseConv2D),
// efficientdet/tf2/efficientdet_keras.py:195-207,459-464,546-556 (depthwise half of SeparableConv2D).
#include <stdlib.h>

#include "common.h"

namespace dwm {

constexpr int THREADS = 256;

template <int CPT> struct Raw;
template <> struct Raw<8> { uint4 v; };
template <> struct Raw<4> { uint2 v; };
template <> struct Raw<2> { uint32_t v; };

template <int CPT> __device__ __forceinline__ Raw<CPT> raw_zero();
template <> __device__ __forceinline__ Raw<8> raw_zero<8>() { Raw<8> r; r.v = make_uint4(0, 0, 0, 0); return r; }
template <> __device__ __forceinline__ Raw<4> raw_zero<4>() { Raw<4> r; r.v = make_uint2(0, 0); return r; }
template <> __device__ __forceinline__ Raw<2> raw_zero<2>() { Raw<2> r; r.v = 0; return r; }

template <int CPT> __device__ __forceinline__ Raw<CPT> raw_load(const bf16_t* p);
template <> __device__ __forceinline__ Raw<8> raw_load<8>(const bf16_t* p) { Raw<8> r; r.v = *reinterpret_cast<const uint4*>(p); return r; }
template <> __device__ __forceinline__ Raw<4> raw_load<4>(const bf16_t* p) { Raw<4> r; r.v = *reinterpret_cast<const uint2*>(p); return r; }
template <> __device__ __forceinline__ Raw<2> raw_load<2>(const bf16_t* p) { Raw<2> r; r.v = *reinterpret_cast<const uint32_t*>(p); return r; }

__device__ __forceinline__ void up2(uint32_t u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}
template <int CPT> __device__ __forceinline__ void raw_unpack(const Raw<CPT>& r, float x[CPT]);
template <> __device__ __forceinline__ void raw_unpack<8>(const Raw<8>& r, float x[8]) {
  up2(r.v.x, x[0], x[1]); up2(r.v.y, x[2], x[3]); up2(r.v.z, x[4], x[5]); up2(r.v.w, x[6], x[7]);
}
template <> __device__ __forceinline__ void raw_unpack<4>(const Raw<4>& r, float x[4]) {
  up2(r.v.x, x[0], x[1]); up2(r.v.y, x[2], x[3]);
}
template <> __device__ __forceinline__ void raw_unpack<2>(const Raw<2>& r, float x[2]) { up2(r.v, x[0], x[1]); }

template <int CPT> __device__ __forceinline__ void store_bf(bf16_t* p, const float x[CPT]);
template <> __device__ __forceinline__ void store_bf<8>(bf16_t* p, const float x[8]) {
  uint4 o;
  o.x = pack2bf(x[0], x[1]); o.y = pack2bf(x[2], x[3]); o.z = pack2bf(x[4], x[5]); o.w = pack2bf(x[6], x[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
template <> __device__ __forceinline__ void store_bf<4>(bf16_t* p, const float x[4]) {
  uint2 o;
  o.x = pack2bf(x[0], x[1]); o.y = pack2bf(x[2], x[3]);
  *reinterpret_cast<uint2*>(p) = o;
}
template <> __device__ __forceinline__ void store_bf<2>(bf16_t* p, const float x[2]) {
  *reinterpret_cast<uint32_t*>(p) = pack2bf(x[0], x[1]);
}

template <int CPT> __device__ __forceinline__ void loadf(const float* p, float x[CPT]) {
#pragma unroll
  for (int e = 0; e < CPT; ++e) x[e] = p[e];
}

// static slot of a (possibly negative) relative output row
__host__ __device__ constexpr int slot_of(int rel, int n) { return ((rel % n) + n) % n; }

struct Args {
  edet_tview_t in;      // fwd / wgrad: activated input view; dgrad: the conv input view (chain target)
  edet_gview_t gy;      // dy (wgrad, dgrad)
  const float* w;       // [K][K][C] fp32
  bf16_t* out; int ldo; // fwd
  float* stat_partials; // fwd
  edet_bwd_epi_t epi;   // dgrad
  float* ws;            // wgrad workspace [P][K*K][C]
  int oh, ow, pad_t, pad_l;
  int nch, ngroups;     // channel chunks (of CPT) per workgroup, channel groups
  int TX, TY;           // tile: TX columns (one per thread) x TY rows of the marched space
  int tiles_x, tiles_y, ntiles, P;
  int xcd;              // 0: block b = (p, g) = (b / ngroups, b % ngroups); 1, 2: XCD-aware maps (lane_setup)
};

struct Lane {
  int chunk, px, g, p, c;
  int tile0, tstep, tend;   // the tiles of this workgroup: tile0, tile0 + tstep, ... < tend
  bool active;
};
template <int CPT>
__device__ __forceinline__ Lane lane_setup(const Args& a, int C) {
  Lane l;
  l.chunk = threadIdx.x % a.nch;
  l.px = threadIdx.x / a.nch;
  l.g = blockIdx.x % a.ngroups;
  l.p = blockIdx.x / a.ngroups;
  l.tile0 = l.p; l.tstep = a.P; l.tend = a.ntiles;
  if (a.xcd) {
    // Block b runs on XCD b % 8 (observed dispatch order; a speed assumption only).  The channel groups of one tile
    // read DIFFERENT parts of the SAME cache lines whenever a pixel's channel row is not a whole number of 128-byte lines
    // (C = 96, 144, 240, 480, 672: r05 counters, fetch = 2.0-2.9 x algorithmic on exactly those layers, 1.15 x at
    // C = 64): with the groups of a tile on different XCDs every L2 fetched every line.  Here the ngroups workgroups
    // of tile slot p sit on ONE XCD, back to back in dispatch order, so the second group hits the first one's lines.
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;   // host: P % 8 == 0
    l.g = j % a.ngroups;
    const int ph = j / a.ngroups;
    l.p = ph * 8 + x;
    l.tile0 = l.p;
    if (a.xcd == 2) {
      // ... and XCD x owns a contiguous range of tiles, its P / 8 resident slots walk it side by side: the column /
      // row halos of neighbouring tiles come from the same L2 too
      const int t8 = (a.ntiles + 7) / 8;
      l.tile0 = x * t8 + ph; l.tstep = a.P / 8; l.tend = min(a.ntiles, (x + 1) * t8);
    }
  }
  l.c = (l.g * a.nch + l.chunk) * CPT;
  l.active = l.px < a.TX && l.c < C;
  return l;
}

// OACT: the view's activation is one of relu / relu6 / hswish (utils.activation_fn; the lite models) -- a template
// parameter of every kernel here, so that the swish / linear instantiations keep their register budget
template <int CPT, bool OACT>
__device__ __forceinline__ void view_act(const edet_tview_t& v, const float sc[CPT], const float sh[CPT],
                                         float x[CPT]) {
  if (v.scale) {
#pragma unroll
    for (int e = 0; e < CPT; ++e) x[e] = fmaf(x[e], sc[e], sh[e]);
  }
  if (OACT) {
#pragma unroll
    for (int e = 0; e < CPT; ++e) x[e] = act_other_(v.act, x[e]);
  } else if (v.act == EDET_ACT_SWISH) {
#pragma unroll
    for (int e = 0; e < CPT; ++e) x[e] = swishf_(x[e]);
  }
}

// workgroup reduction of per-thread channel sums into one partial row: partials[(p*nrow + row)*C + c].
// One round per row: every thread parks its CPT values in LDS ([px][chunk][e] = thread-major), then
// thread (col, slice) sums the pixels px = slice, slice + nsl, ... of column col into its own slot and the
// few slices of a column are added in slice order (r04: no LDS atomics -- the same sums on every run).
template <int CPT, int NROW>
__device__ __forceinline__ void block_channel_sums(const Args& a, const Lane& l, int C, const float (&s)[NROW][CPT],
                                                   float* dst_rows, float* red /* LDS [THREADS*CPT + width] */) {
  const int width = a.nch * CPT;
  const int nthr = a.TX * a.nch;                 // threads that own a (pixel, chunk)
  const int nsl = THREADS / width;               // pixel slices per column
  const int col = threadIdx.x % width, slice = threadIdx.x / width;
  float* out = red + THREADS * CPT;
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CPT; ++e) red[threadIdx.x * CPT + e] = (l.active && threadIdx.x < nthr) ? s[r][e] : 0.f;
    __syncthreads();
    if (slice < nsl) {
      float t = 0.f;
      for (int px = slice; px < a.TX; px += nsl) t += red[px * width + col];
      out[slice * width + col] = t;
    }
    __syncthreads();
    if (threadIdx.x < width) {
      const int c = l.g * width + threadIdx.x;
      float t = 0.f;
      for (int q = 0; q < nsl; ++q) t += out[q * width + threadIdx.x];
      if (c < C) dst_rows[((size_t)l.p * NROW + r) * C + c] = t;
    }
  }
}

// =====================================================================================================
// LDS row-exchange variants (the default).  PMC counters of the kernels above showed them 50-76 % VALU-bound,
// not HBM-bound: every thread re-applied the producer's BatchNorm + swish (or the BatchNorm backward) to
// each of the K horizontally adjacent pixels it loaded, i.e. K times per element.  Here a thread loads and
// transforms only ITS OWN pixel of a row (plus one halo pixel for the first K - S threads), parks the
// fp32 result in a two-row LDS ring, and after one workgroup barrier per row reads the K neighbours back
// from LDS (conflict-free: consecutive lanes = consecutive channels).  Global loads stay two rows ahead of
// the barrier, so the HBM latency is still covered by the march; VALU work per element drops from
// K*(act) + K*K to act + K*K and the L1 traffic from K to 1 load per element.
template <int CPT> __device__ __forceinline__ void lds_put(float* p, const float x[CPT]);
template <> __device__ __forceinline__ void lds_put<4>(float* p, const float x[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
}
template <> __device__ __forceinline__ void lds_put<2>(float* p, const float x[2]) {
  *reinterpret_cast<float2*>(p) = make_float2(x[0], x[1]);
}
template <> __device__ __forceinline__ void lds_put<8>(float* p, const float x[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
}
template <int CPT> __device__ __forceinline__ void lds_get(const float* p, float x[CPT]);
template <> __device__ __forceinline__ void lds_get<8>(const float* p, float x[8]) {
  const float4 u = *reinterpret_cast<const float4*>(p);
  const float4 v = *reinterpret_cast<const float4*>(p + 4);
  x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
}
template <> __device__ __forceinline__ void lds_get<4>(const float* p, float x[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}
template <> __device__ __forceinline__ void lds_get<2>(const float* p, float x[2]) {
  const float2 v = *reinterpret_cast<const float2*>(p);
  x[0] = v.x; x[1] = v.y;
}

// floats of LDS in front of the row ring (block_channel_sums scratch)
__host__ __device__ inline int red_floats(int nch, int cpt) { return THREADS * cpt + THREADS; }

// PF = rows of global loads in flight per thread ahead of the row being consumed (register FIFO).  With two
// rows the march was latency-bound: a row step (~0.3 us of work) had to wait for a load issued only two
// steps earlier (HBM latency under load ~2 us).
// r02j lab: deeper forward FIFOs (8-9 rows at stride 1, 6 at stride 2) are slower (4.79 -> 5.24 ms over the 15 layer
// shapes: they cost the fourth wave per SIMD), and so are stride-2 gradient kernels with 3-6 rows in flight at two
// waves per SIMD (11.05 -> 11.29 ms).
template <int S, int CPT> struct PfDepth {
  static constexpr int fwd = CPT == 8 ? 3 : (S == 1 ? 6 : 4);
  static constexpr int bwd = CPT == 4 ? (S == 1 ? 4 : 2) : (S == 1 ? 6 : 3);   // register budget of the 3-wave kernels
  static constexpr int dgrad = CPT == 4 ? (S == 1 ? 3 : 1) : (S == 1 ? 6 : 3);
};

__host__ __device__ constexpr int gcd_(int x, int y) { return y == 0 ? x : gcd_(y, x % y); }

// weight gradient with the activated input row exchanged through LDS (see k_fwd_v2); dy is the thread's own
// column, so its BatchNorm backward was already applied once per element.
template <int K, int S, int CPT, bool GBN, bool OACT>
__global__ __launch_bounds__(THREADS) void k_wgrad_lx(const Args a) {
  constexpr int PF = PfDepth<S, CPT>::bwd;          // input rows in flight
  constexpr int PG = (PF + S - 1) / S + 1;     // dy rows in flight
  constexpr int NSL = (K + S - 1) / S;
  constexpr int U = S * NSL;
  constexpr int HALO = K - S;
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX * S + HALO;
  float* ring = red + red_floats(a.nch, CPT);
  const bool in_tile = l.px < a.TX;
  const bf16_t* IN = reinterpret_cast<const bf16_t*>(a.in.data);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gy.dz);
  const bf16_t* YY = reinterpret_cast<const bf16_t*>(a.gy.y);
  float wacc[K * K][CPT], sc[CPT], sh[CPT], ga[CPT], gb[CPT], gc[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) { sc[e] = 1.f; sh[e] = 0.f; ga[e] = 1.f; gb[e] = 0.f; gc[e] = 0.f; }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < CPT; ++e) wacc[t][e] = 0.f;
  if (l.active) {
    if (a.in.scale) { loadf<CPT>(a.in.scale + l.c, sc); loadf<CPT>(a.in.shift + l.c, sh); }
    if (GBN) { loadf<CPT>(a.gy.a + l.c, ga); loadf<CPT>(a.gy.b + l.c, gb); loadf<CPT>(a.gy.cc + l.c, gc); }
  }

  for (int tile = l.tile0; tile < l.tend; tile += l.tstep) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty = rr / a.tiles_x, tx = rr - ty * a.tiles_x;
    const int oy0 = ty * a.TY, oy1 = min(a.oh, oy0 + a.TY);
    const int ox = tx * a.TX + l.px;
    const bool xok = l.active && ox < a.ow;
    const int wc0 = tx * a.TX * S - a.pad_l;
    bool mok[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int ix = wc0 + l.px * S + j;
      mok[j] = l.active && ix >= 0 && ix < W;
    }
    const int ixh = wc0 + a.TX * S + l.px;
    const bool hown = HALO > 0 && l.px < HALO && in_tile;
    const bool hok = hown && l.c < C && ixh >= 0 && ixh < W;
    const bf16_t* ibase = IN + ((int64_t)n * H * W * a.in.ld + l.c);
    const size_t gbase = ((size_t)n * a.oh * a.ow + ox) * a.gy.ld + l.c;
    float dyw[NSL][CPT];                       // dy rows in flight, slot = oy mod NSL
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int e = 0; e < CPT; ++e) dyw[s][e] = 0.f;

    const int t0 = (oy0 * S / U) * U, t_last = (oy1 - 1) * S + K - 1;
    Raw<CPT> fm[PF + 1][S], fh[PF + 1], fz[PG + 1], fy[GBN ? PG + 1 : 1];
#pragma unroll
    for (int i = 0; i <= PF; ++i) {
      fh[i] = raw_zero<CPT>();
#pragma unroll
      for (int j = 0; j < S; ++j) fm[i][j] = raw_zero<CPT>();
    }
#pragma unroll
    for (int i = 0; i <= PG; ++i) {
      fz[i] = raw_zero<CPT>();
      if (GBN) fy[i] = raw_zero<CPT>();
    }
    auto load_row = [&](int t, Raw<CPT> (&dst)[S], Raw<CPT>& hdst) {
      const int r = t - a.pad_t;
      if (t <= t_last && r >= 0 && r < H) {
        const bf16_t* rp = ibase + (int64_t)r * W * a.in.ld;
#pragma unroll
        for (int j = 0; j < S; ++j)
          dst[j] = mok[j] ? raw_load<CPT>(rp + (int64_t)(wc0 + l.px * S + j) * a.in.ld) : raw_zero<CPT>();
        if (HALO > 0) hdst = hok ? raw_load<CPT>(rp + (int64_t)ixh * a.in.ld) : raw_zero<CPT>();
      }
    };
    auto load_dy = [&](int oy, Raw<CPT>& gz, Raw<CPT>& gyr) {   // raw dy row oy (zero outside the tile / image)
      gz = raw_zero<CPT>();
      if (GBN) gyr = raw_zero<CPT>();
      if (oy >= oy0 && oy < oy1 && xok) {
        const size_t off = gbase + (size_t)oy * a.ow * a.gy.ld;
        gz = raw_load<CPT>(DZ + off);
        if (GBN) gyr = raw_load<CPT>(YY + off);
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(t0 + i, fm[i], fh[i]);
#pragma unroll
    for (int i = 0; i < PG; ++i) load_dy(t0 / S + i, fz[i], fy[GBN ? i : 0]);
    for (int tb = t0; tb <= t_last; tb += U) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int t = tb + tt;
        load_row(t + PF, fm[PF], fh[PF]);
        if (tt % S == 0) {                     // static: dy row oy = t / S enters the window at ky = 0
          const int sl = slot_of(tt / S, NSL);
          const int oy = t / S;
          load_dy(oy + PG, fz[PG], fy[GBN ? PG : 0]);
          float g[CPT];
          raw_unpack<CPT>(fz[0], g);
          if (GBN) {
            float y[CPT];
            raw_unpack<CPT>(fy[0], y);
            const bool in_t = oy >= oy0 && oy < oy1 && xok;
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[e] = in_t ? fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e])) : 0.f;
          }
#pragma unroll
          for (int e = 0; e < CPT; ++e) dyw[sl][e] = g[e];
#pragma unroll
          for (int i = 0; i < PG; ++i) {
            fz[i] = fz[i + 1];
            if (GBN) fy[i] = fy[i + 1];
          }
        }
        const int r = t - a.pad_t;
        const bool row_ok = t <= t_last && r >= 0 && r < H;
        float* buf = ring + (t & 1) * WIN * width;
        if (row_ok && in_tile) {
#pragma unroll
          for (int j = 0; j < S; ++j) {
            float x[CPT];
            raw_unpack<CPT>(fm[0][j], x);
            view_act<CPT, OACT>(a.in, sc, sh, x);
            if (!mok[j]) {
#pragma unroll
              for (int e = 0; e < CPT; ++e) x[e] = 0.f;
            }
            lds_put<CPT>(buf + (l.px * S + j) * width + l.chunk * CPT, x);
          }
          if (hown) {
            float x[CPT];
            raw_unpack<CPT>(fh[0], x);
            view_act<CPT, OACT>(a.in, sc, sh, x);
            if (!hok) {
#pragma unroll
              for (int e = 0; e < CPT; ++e) x[e] = 0.f;
            }
            lds_put<CPT>(buf + (a.TX * S + l.px) * width + l.chunk * CPT, x);
          }
        }
        __syncthreads();
        if (row_ok && in_tile) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            float x[CPT];
            lds_get<CPT>(buf + (l.px * S + kx) * width + l.chunk * CPT, x);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
              if ((tt - ky) % S == 0) {
                const int sl = slot_of((tt - ky) / S, NSL);
#pragma unroll
                for (int e = 0; e < CPT; ++e) wacc[ky * K + kx][e] = fmaf(x[e], dyw[sl][e], wacc[ky * K + kx][e]);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          fh[i] = fh[i + 1];
#pragma unroll
          for (int j = 0; j < S; ++j) fm[i][j] = fm[i + 1][j];
        }
      }
    }
    __syncthreads();
  }
  block_channel_sums<CPT, K * K>(a, l, C, wacc, a.ws, red);
}

// data gradient with the BatchNorm-backward-transformed dy row exchanged through LDS: thread q loads and
// transforms dy[oy][q] (plus the halo columns q0 - (D-1) .. q0 - 1 by the first D - 1 threads) and reads
// dy[oy][q - d] (d < D) back after the barrier.  Window column wc <-> dy column q0 - (D-1) + wc.
template <int K, int S, int CPT, bool GBN, bool OACT>
__global__ __launch_bounds__(THREADS, 3) void k_dgrad_lx(const Args a) {
  constexpr int PF = PfDepth<S, CPT>::dgrad;       // dy rows (and the saved-input rows they complete) in flight
  constexpr int D = (K + S - 1) / S;
  constexpr int RS = S * D;
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX + D - 1;
  float* ring = red + red_floats(a.nch, CPT);
  const bool in_tile = l.px < a.TX;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.in.data);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gy.dz);
  const bf16_t* YY = reinterpret_cast<const bf16_t*>(a.gy.y);
  bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);
  float w[K * K][CPT], sc[CPT], sh[CPT], ga[CPT], gb[CPT], gc[CPT], mu[CPT], rs[CPT];
  float st[2][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) {
    sc[e] = 1.f; sh[e] = 0.f; ga[e] = 1.f; gb[e] = 0.f; gc[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f;
    st[0][e] = st[1][e] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < CPT; ++e) w[t][e] = l.active ? a.w[(size_t)t * C + l.c + e] : 0.f;
  const bool want_stats = a.epi.stat_partials != nullptr;
  constexpr bool other = OACT;
  const bool swish = !OACT && a.in.act == EDET_ACT_SWISH;
  if (l.active) {
    if (a.in.scale) { loadf<CPT>(a.in.scale + l.c, sc); loadf<CPT>(a.in.shift + l.c, sh); }
    if (GBN) { loadf<CPT>(a.gy.a + l.c, ga); loadf<CPT>(a.gy.b + l.c, gb); loadf<CPT>(a.gy.cc + l.c, gc); }
    if (want_stats) { loadf<CPT>(a.epi.mean + l.c, mu); loadf<CPT>(a.epi.rstd + l.c, rs); }
  }

  const int QW = (W + a.pad_l + S - 1) / S, QH = (H + a.pad_t + S - 1) / S;
  for (int tile = l.tile0; tile < l.tend; tile += l.tstep) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty_ = rr / a.tiles_x, tx_ = rr - ty_ * a.tiles_x;
    const int q0 = tx_ * a.TX;
    const int q = q0 + l.px;
    const bool qok = l.active && q < QW;
    const int qy0 = ty_ * a.TY, qy1 = min(QH, qy0 + a.TY);
    const bool mok = l.active && in_tile && q < a.ow;               // own dy column q (>= 0 always)
    const int qh = q0 - (D - 1) + l.px;                            // halo dy column
    const bool hown = D > 1 && l.px < D - 1 && in_tile;
    const bool hok = hown && l.c < C && qh >= 0 && qh < a.ow;
    float acc[RS][S][CPT];
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
      for (int u = 0; u < S; ++u)
#pragma unroll
        for (int e = 0; e < CPT; ++e) acc[s][u][e] = 0.f;
    const size_t gimg = (size_t)n * a.oh * a.ow;
    const int o_begin = ((qy0 - (D - 1)) >= 0 ? (qy0 - (D - 1)) / D : -((D - 1 - (qy0 - (D - 1))) / D)) * D;
    Raw<CPT> fz[PF + 1], fy[GBN ? PF + 1 : 1], fhz[PF + 1], fhy[GBN ? PF + 1 : 1], fx[PF + 1][S][S];
#pragma unroll
    for (int i = 0; i <= PF; ++i) {
      fz[i] = fhz[i] = raw_zero<CPT>();
      if (GBN) fy[i] = fhy[i] = raw_zero<CPT>();
#pragma unroll
      for (int v = 0; v < S; ++v)
#pragma unroll
        for (int u = 0; u < S; ++u) fx[i][v][u] = raw_zero<CPT>();
    }
    if (!GBN) fy[0] = fhy[0] = raw_zero<CPT>();
    const bool need_x = swish || other || want_stats;
    // saved conv input of the S x S pixels that dy step oy completes (needed for act' / BN backward sums)
    auto load_x = [&](int oy, Raw<CPT> (&xr)[S][S]) {
#pragma unroll
      for (int v = 0; v < S; ++v) {
        const int iy = oy * S + v - a.pad_t;
#pragma unroll
        for (int u = 0; u < S; ++u) {
          const int ix = q * S + u - a.pad_l;
          xr[v][u] = raw_zero<CPT>();
          if (need_x && oy >= qy0 && oy < qy1 && iy >= 0 && iy < H && qok && ix >= 0 && ix < W)
            xr[v][u] = raw_load<CPT>(X + ((size_t)(n * H + iy) * W + ix) * a.in.ld + l.c);
        }
      }
    };
    auto load_dyrow = [&](int oy, Raw<CPT>& z, Raw<CPT>& y, Raw<CPT>& hz, Raw<CPT>& hy) {
      if (oy < qy1 && oy >= 0 && oy < a.oh) {             // uniform
        const size_t rowoff = (gimg + (size_t)oy * a.ow) * a.gy.ld + l.c;
        z = raw_zero<CPT>();
        if (GBN) y = raw_zero<CPT>();
        if (mok) {
          z = raw_load<CPT>(DZ + rowoff + (size_t)q * a.gy.ld);
          if (GBN) y = raw_load<CPT>(YY + rowoff + (size_t)q * a.gy.ld);
        }
        if (D > 1) {
          hz = raw_zero<CPT>();
          if (GBN) hy = raw_zero<CPT>();
          if (hok) {
            hz = raw_load<CPT>(DZ + rowoff + (size_t)qh * a.gy.ld);
            if (GBN) hy = raw_load<CPT>(YY + rowoff + (size_t)qh * a.gy.ld);
          }
        }
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      load_dyrow(o_begin + i, fz[i], fy[GBN ? i : 0], fhz[i], fhy[GBN ? i : 0]);
      load_x(o_begin + i, fx[i]);
    }
    for (int ob = o_begin; ob < qy1; ob += D) {
#pragma unroll
      for (int oo = 0; oo < D; ++oo) {
        const int oy = ob + oo;
        load_dyrow(oy + PF, fz[PF], fy[GBN ? PF : 0], fhz[PF], fhy[GBN ? PF : 0]);   // PF rows in flight
        load_x(oy + PF, fx[PF]);
        Raw<CPT> (&xr)[S][S] = fx[0];
        const Raw<CPT> cz = fz[0], cy = fy[0], hcz = fhz[0], hcy = fhy[0];
        const bool row_ok = oy < qy1 && oy >= 0 && oy < a.oh;   // uniform
        float* buf = ring + (oy & 1) * WIN * width;
        if (row_ok && in_tile) {
          float g[CPT];
          raw_unpack<CPT>(cz, g);
          if (GBN) {
            float y[CPT];
            raw_unpack<CPT>(cy, y);
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[e] = mok ? fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e])) : 0.f;
          }
          lds_put<CPT>(buf + (l.px + D - 1) * width + l.chunk * CPT, g);
          if (hown) {
            raw_unpack<CPT>(hcz, g);
            if (GBN) {
              float y[CPT];
              raw_unpack<CPT>(hcy, y);
#pragma unroll
              for (int e = 0; e < CPT; ++e) g[e] = hok ? fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e])) : 0.f;
            }
            lds_put<CPT>(buf + l.px * width + l.chunk * CPT, g);
          }
        }
        __syncthreads();
        if (row_ok && in_tile) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            float g[CPT];
            lds_get<CPT>(buf + (l.px + D - 1 - d) * width + l.chunk * CPT, g);
            // dy[oy][q-d] feeds tx = S*q + u with kx = u + S*d, and ty = oy*S + ky
#pragma unroll
            for (int u = 0; u < S; ++u) {
              if (u + S * d < K) {
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                  const int sl = slot_of(oo * S + ky, RS);     // (oy*S + ky) mod RS, ob*S = 0 mod RS
#pragma unroll
                  for (int e = 0; e < CPT; ++e)
                    acc[sl][u][e] = fmaf(w[ky * K + u + S * d][e], g[e], acc[sl][u][e]);
                }
              }
            }
          }
        }
        // rows ty = oy*S + v (v < S) are complete
#pragma unroll
        for (int v = 0; v < S; ++v) {
          const int sl = slot_of(oo * S + v, RS);
          const int iy = oy * S + v - a.pad_t;
          if (oy >= qy0 && oy < qy1 && iy >= 0 && iy < H) {   // uniform
#pragma unroll
            for (int u = 0; u < S; ++u) {
              const int ix = q * S + u - a.pad_l;
              if (qok && ix >= 0 && ix < W) {
                const size_t off = ((size_t)(n * H + iy) * W + ix) * a.in.ld + l.c;
                float g[CPT], x[CPT];
#pragma unroll
                for (int e = 0; e < CPT; ++e) g[e] = acc[sl][u][e];
                raw_unpack<CPT>(xr[v][u], x);
                if (swish) {
#pragma unroll
                  for (int e = 0; e < CPT; ++e) g[e] *= swish_gradf_(fmaf(x[e], sc[e], sh[e]));
                } else if (other) {
#pragma unroll
                  for (int e = 0; e < CPT; ++e) g[e] *= act_other_grad_(a.in.act, fmaf(x[e], sc[e], sh[e]));
                }
                if (a.epi.beta) {
                  float old[CPT];
                  raw_unpack<CPT>(raw_load<CPT>(GO + off), old);
#pragma unroll
                  for (int e = 0; e < CPT; ++e) g[e] += old[e];
                }
                store_bf<CPT>(GO + off, g);
                if (want_stats) {
#pragma unroll
                  for (int e = 0; e < CPT; ++e) {
                    st[0][e] += g[e];
                    st[1][e] = fmaf(g[e], (x[e] - mu[e]) * rs[e], st[1][e]);
                  }
                }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < S; ++u)
#pragma unroll
            for (int e = 0; e < CPT; ++e) acc[sl][u][e] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          fz[i] = fz[i + 1];
          fhz[i] = fhz[i + 1];
          if (GBN) { fy[i] = fy[i + 1]; fhy[i] = fhy[i + 1]; }
#pragma unroll
          for (int v = 0; v < S; ++v)
#pragma unroll
            for (int u = 0; u < S; ++u) fx[i][v][u] = fx[i + 1][v][u];
        }
      }
    }
    __syncthreads();
  }
  if (want_stats) block_channel_sums<CPT, 2>(a, l, C, st, a.epi.stat_partials, red);
}

// =====================================================================================================
// One-pass backward for ANY stride (round 6): data gradient AND weight gradient from a single read of (dz, y, x).
// Round 5 ran the stride-2 layers as k_dgrad_lx + k_wgrad_lx -- two marches over the same three tensors (320x320x96:
// 7.6 GB fetched by each) -- and the stride-1 fused kernel above spent two thirds of its VALU slots outside the
// multiply-adds (r06 ISA count of k_bwd_fused<5, 2>: 172 VALU instructions per row step for 50 packed FMAs: 64-bit
// address arithmetic per load, 38 register moves feeding badly paired v_pk_fma_f32, cndmask chains).  This kernel:
//   * thread = dy column q x CPT channels, marches over the dy rows oy (as k_dgrad_lx): the transformed dy row goes
//     through the LDS ring, the D = ceil(K / S) column neighbours dy[oy][q - d] come back after one barrier;
//   * the thread's OWN S input columns ix = q S + u - pad_l are kept as a K-row register window of act(z) (weight
//     gradient), act'(z) and the raw x (epilogue): S new input rows enter per step, the S rows that no later dy row
//     touches leave through the epilogue (act', accumulate, store, BatchNorm-backward sums);
//   * every neighbour value g feeds both accumulations:
//       dacc[row oy S + ky][u] += w[ky][u + S d] g        wacc[ky][u + S d] += act(z)[row oy S + ky][u] g
//   * all arithmetic on explicit 2-vectors of adjacent channels (v_pk_fma_f32 with naturally paired registers);
//   * every global address = uniform row pointer (SGPR pair) + a per-thread 32-bit byte offset that is constant for
//     the whole tile: no per-load VALU address arithmetic.
// Window slots are rows modulo K (a row leaves in the step before the row K later enters); the row loop is unrolled
// over lcm(K, NF) steps so that slots and the load FIFO are static register indices.  x pixels outside the tile enter
// the window as zeros: every (x pixel, dy pixel) pair is counted by exactly one tile.
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NV> struct RawV { uint32_t u[NV]; };
// Raw buffer access: address = descriptor base (4 SGPRs, uniform: the image) + soff (SGPR, uniform: the row) + voff (one
// VGPR, the thread's column / channel offset, constant over a tile) -- no VALU address arithmetic per load.  (Plain
// pointers did not get there: the compiler widened the hoisted 32-bit offsets to 64-bit VGPR pairs and added the row
// pointer on the VALU, two to four instructions per load.)  num_records = 2^31 - 1: the range check is not used,
// rows and columns are clamped by the caller.
typedef __amdgpu_buffer_rsrc_t brsrc_t;
__device__ __forceinline__ brsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <int NV> __device__ __forceinline__ RawV<NV> ldg(brsrc_t r, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ RawV<1> ldg<1>(brsrc_t r, uint32_t voff, uint32_t soff) {
  RawV<1> v; v.u[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); return v;
}
template <> __device__ __forceinline__ RawV<2> ldg<2>(brsrc_t r, uint32_t voff, uint32_t soff) {
  const auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  RawV<2> v; v.u[0] = t[0]; v.u[1] = t[1]; return v;
}
template <int NV> __device__ __forceinline__ void unpackv(const RawV<NV>& r, f2 (&x)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    x[i].x = __uint_as_float(r.u[i] << 16);
    x[i].y = __uint_as_float(r.u[i] & 0xffff0000u);
  }
}
template <int NV> __device__ __forceinline__ void stg(brsrc_t r, uint32_t voff, uint32_t soff, const f2 (&x)[NV]);
template <> __device__ __forceinline__ void stg<1>(brsrc_t r, uint32_t voff, uint32_t soff, const f2 (&x)[1]) {
  __builtin_amdgcn_raw_buffer_store_b32(pack2bf(x[0].x, x[0].y), r, voff, soff, 0);
}
template <> __device__ __forceinline__ void stg<2>(brsrc_t r, uint32_t voff, uint32_t soff, const f2 (&x)[2]) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  u2 o; o[0] = pack2bf(x[0].x, x[0].y); o[1] = pack2bf(x[1].x, x[1].y);
  __builtin_amdgcn_raw_buffer_store_b64(o, r, voff, soff, 0);
}
template <int NV> __device__ __forceinline__ void lds_putv(float* p, const f2 (&x)[NV]);
template <> __device__ __forceinline__ void lds_putv<1>(float* p, const f2 (&x)[1]) { *reinterpret_cast<f2*>(p) = x[0]; }
template <> __device__ __forceinline__ void lds_putv<2>(float* p, const f2 (&x)[2]) {
  *reinterpret_cast<float4*>(p) = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
}
template <int NV> __device__ __forceinline__ void lds_getv(const float* p, f2 (&x)[NV]);
template <> __device__ __forceinline__ void lds_getv<1>(const float* p, f2 (&x)[1]) { x[0] = *reinterpret_cast<const f2*>(p); }
template <> __device__ __forceinline__ void lds_getv<2>(const float* p, f2 (&x)[2]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  x[0].x = v.x; x[0].y = v.y; x[1].x = v.z; x[1].y = v.w;
}
template <int NV> __device__ __forceinline__ void loadv(const float* p, f2 (&x)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) x[i] = *reinterpret_cast<const f2*>(p + 2 * i);
}

// load FIFO depth (register sets) of k_bwd_one.  r06 lab: twice the depth changes nothing (7.97 -> 7.87 ms over the 15
// layer shapes) -- these kernels are bound by instruction issue at two waves per SIMD, not by bytes in flight.
template <int K, int S, int CPT> struct OneDepth {
  static constexpr int nf = K == 3 ? ((CPT == 4 || S == 2) ? 3 : 6) : 5;
};

// floor(a / b), b > 0
__host__ __device__ constexpr int fdiv_(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__host__ __device__ constexpr int cdiv_(int a, int b) { return -fdiv_(-a, b); }
// a uniform "lo <= v < lo + span" as one unsigned compare
struct URange {
  int lo; uint32_t span;
  __device__ __forceinline__ void set(int l, int h) { lo = l; span = h > l ? (uint32_t)(h - l) : 0u; }
  __device__ __forceinline__ bool has(int v) const { return (uint32_t)(v - lo) < span; }
};

// ACTM: activation of the input view -- 0 none (a stored tensor), 1 swish, 2 relu / relu6 / hswish / mish / srelu
template <int K, int S, int CPT, bool GBN, int ACTM>
__global__ __launch_bounds__(THREADS, (CPT * K >= 10 || S == 2) ? 2 : 3) void k_bwd_one(const Args a) {
  constexpr int NV = CPT / 2;
  constexpr int D = (K + S - 1) / S;
  constexpr int NF = OneDepth<K, S, CPT>::nf;
  constexpr int PF = NF - 1;
  constexpr int U = K * NF / gcd_(K, NF);
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX + D - 1;
  float* ring = red + red_floats(a.nch, CPT);
  const bool in_tile = l.px < a.TX;
  const char* X = reinterpret_cast<const char*>(a.in.data);
  const char* DZ = reinterpret_cast<const char*>(a.gy.dz);
  const char* YY = reinterpret_cast<const char*>(a.gy.y);
  char* GO = reinterpret_cast<char*>(a.epi.gout);
  f2 w[K * K][NV], wacc[K * K][NV], sc[NV], sh[NV], ga[NV], gb[NV], gc[NV], mu[NV], rs[NV], st[2][NV];
  const f2 zero2 = {0.f, 0.f}, one2 = {1.f, 1.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    sc[i] = one2; sh[i] = zero2; ga[i] = one2; gb[i] = zero2; gc[i] = zero2; mu[i] = zero2; rs[i] = one2;
    st[0][i] = st[1][i] = zero2;
  }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      w[t][i] = zero2;
      if (l.active) w[t][i] = *reinterpret_cast<const f2*>(a.w + (size_t)t * C + l.c + 2 * i);
      wacc[t][i] = zero2;
    }
  const bool want_stats = a.epi.stat_partials != nullptr;
  if (l.active) {
    if (a.in.scale) { loadv<NV>(a.in.scale + l.c, sc); loadv<NV>(a.in.shift + l.c, sh); }
    if (GBN) { loadv<NV>(a.gy.a + l.c, ga); loadv<NV>(a.gy.b + l.c, gb); loadv<NV>(a.gy.cc + l.c, gc); }
    if (want_stats) { loadv<NV>(a.epi.mean + l.c, mu); loadv<NV>(a.epi.rstd + l.c, rs); }
  }
  const uint32_t cc2 = (uint32_t)(l.c < C ? l.c : 0) * 2u;
  const uint32_t xrow_b = (uint32_t)W * a.in.ld * 2, grow_b = (uint32_t)a.ow * a.gy.ld * 2;   // < 2^31 per image (host check)
  const uint32_t lown = (uint32_t)((l.px + D - 1) * width + l.chunk * CPT);   // ring slot of the own dy column (floats)
  const uint32_t lhalo = (uint32_t)(l.px * width + l.chunk * CPT);

  const int QW = (W + a.pad_l + S - 1) / S, QH = (H + a.pad_t + S - 1) / S;
  for (int tile = l.tile0; tile < l.tend; tile += l.tstep) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty_ = rr / a.tiles_x, tx_ = rr - ty_ * a.tiles_x;
    const int q0 = tx_ * a.TX;
    const int q = q0 + l.px;
    const int qy0 = ty_ * a.TY, qy1 = min(QH, qy0 + a.TY);
    const bool mok = l.active && q < a.ow;                             // own dy column
    const int qh = q0 - (D - 1) + l.px;                                // halo dy column (first D - 1 pixel lanes)
    const bool hown = D > 1 && l.px < D - 1 && in_tile;
    const bool hok = hown && l.c < C && qh >= 0 && qh < a.ow;
    bool cok[S];
    uint32_t xoff[S];
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const int ix = q * S + u - a.pad_l;
      cok[u] = l.active && q < QW && ix >= 0 && ix < W;
      xoff[u] = (uint32_t)min(max(ix, 0), W - 1) * (uint32_t)(a.in.ld * 2) + cc2;
    }
    const uint32_t goff = (uint32_t)min(q, a.ow - 1) * (uint32_t)(a.gy.ld * 2) + cc2;
    const uint32_t hoff = hown ? (uint32_t)min(max(qh, 0), a.ow - 1) * (uint32_t)(a.gy.ld * 2) + cc2 : goff;
    const brsrc_t x_img = make_rsrc(X + (size_t)n * H * xrow_b);
    const brsrc_t z_img = make_rsrc(DZ + (size_t)n * a.oh * grow_b);
    const brsrc_t y_img = make_rsrc(GBN ? YY + (size_t)n * a.oh * grow_b : DZ);
    const brsrc_t go_img = make_rsrc(GO + (size_t)n * H * xrow_b);

    // Uniform row conditions as ranges of the dy row index oy (one unsigned compare each):
    //   dy row oy is read:                 0 <= oy < oh, o_first <= oy < qy1
    //   entering x row j (rho = oy S + K - S + j, r = rho - pad_t) lies in the image and belongs to this tile
    //   (rho / S = oy + (K - S + j) / S in [qy0, qy1))
    //   completing row v (r = oy S + v - pad_t) lies in the image, oy in [qy0, qy1)
    const int o_first = qy0 - (D - 1);          // first dy row that feeds a row of this tile; window slots are relative to it
    URange dy_rng, x_rng[S], c_rng[S];
    dy_rng.set(max(0, o_first), min(a.oh, qy1));
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int e = (K - S + j) / S, c = K - S + j - a.pad_t;
      x_rng[j].set(max(qy0 - e, cdiv_(-c, S)), min(qy1 - e, cdiv_(H - c, S)));
      c_rng[j].set(max(qy0, cdiv_(a.pad_t - j, S)), min(qy1, cdiv_(H + a.pad_t - j, S)));
    }

    f2 dacc[K][S][NV], xt[K][S][NV], dsw[K][S][NV];
    RawV<NV> xw[K][S];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int u = 0; u < S; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i) { dacc[s][u][i] = zero2; xt[s][u][i] = zero2; dsw[s][u][i] = one2; xw[s][u].u[i] = 0; }

    RawV<NV> fz[NF], fy[GBN ? NF : 1], fhz[NF], fhy[GBN ? NF : 1], fx[NF][S][S];
    // every load of a step is issued on every step at an in-range address (see k_fwd_v2)
    auto load_step = [&](int oy, RawV<NV>& z, RawV<NV>& y, RawV<NV>& hz, RawV<NV>& hy, RawV<NV> (&x)[S][S]) {
      const uint32_t zr = (uint32_t)min(max(oy, 0), a.oh - 1) * grow_b;
      z = ldg<NV>(z_img, goff, zr);
      if (D > 1) hz = ldg<NV>(z_img, hoff, zr);
      if (GBN) {
        y = ldg<NV>(y_img, goff, zr);
        if (D > 1) hy = ldg<NV>(y_img, hoff, zr);
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const uint32_t xr = (uint32_t)min(max(oy * S + K - S + j - a.pad_t, 0), H - 1) * xrow_b;
#pragma unroll
        for (int u = 0; u < S; ++u) x[j][u] = ldg<NV>(x_img, xoff[u], xr);
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_step(o_first + i, fz[i], fy[GBN ? i : 0], fhz[i], fhy[GBN ? i : 0], fx[i]);
    for (int ob = o_first; ob < qy1; ob += U) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int oy = ob + tt;
        constexpr int dummy_ = 0; (void)dummy_;
        const int fc = tt % NF, fn = (tt + PF) % NF;
        load_step(oy + PF, fz[fn], fy[GBN ? fn : 0], fhz[fn], fhy[GBN ? fn : 0], fx[fn]);
        // ---- S new input rows enter the window
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const int sl = (tt * S + K - S + j) % K;
          const bool rin = x_rng[j].has(oy);                            // uniform
#pragma unroll
          for (int u = 0; u < S; ++u) {
            xw[sl][u] = fx[fc][j][u];
            if (rin) {
              f2 x[NV];
              unpackv<NV>(fx[fc][j][u], x);
#pragma unroll
              for (int i = 0; i < NV; ++i) {
                const f2 z = x[i] * sc[i] + sh[i];
                f2 v, dv;
                if (ACTM == 1) {
                  f2 sg;
                  sg.x = sigmoidf_(z.x); sg.y = sigmoidf_(z.y);
                  v = z * sg;
                  dv = sg * (one2 + z * (one2 - sg));
                } else if (ACTM == 2) {
                  v.x = act_other_(a.in.act, z.x); v.y = act_other_(a.in.act, z.y);
                  dv.x = act_other_grad_(a.in.act, z.x); dv.y = act_other_grad_(a.in.act, z.y);
                } else {
                  v = z; dv = one2;
                }
                xt[sl][u][i] = cok[u] ? v : zero2;
                if (ACTM != 0) dsw[sl][u][i] = dv;
              }
            } else {
#pragma unroll
              for (int i = 0; i < NV; ++i) xt[sl][u][i] = zero2;
            }
          }
        }
        // ---- dy row oy through the LDS ring
        const bool row_ok = dy_rng.has(oy);                             // uniform
        float* buf = ring + (oy & 1) * WIN * width;
        if (row_ok && in_tile) {
          f2 g[NV];
          unpackv<NV>(fz[fc], g);
          if (GBN) {
            f2 y[NV];
            unpackv<NV>(fy[GBN ? fc : 0], y);
#pragma unroll
            for (int i = 0; i < NV; ++i) g[i] = ga[i] * g[i] + (gb[i] * y[i] + gc[i]);
          }
#pragma unroll
          for (int i = 0; i < NV; ++i) g[i] = mok ? g[i] : zero2;
          lds_putv<NV>(buf + lown, g);
          if (hown) {
            unpackv<NV>(fhz[fc], g);
            if (GBN) {
              f2 y[NV];
              unpackv<NV>(fhy[GBN ? fc : 0], y);
#pragma unroll
              for (int i = 0; i < NV; ++i) g[i] = ga[i] * g[i] + (gb[i] * y[i] + gc[i]);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) g[i] = hok ? g[i] : zero2;
            lds_putv<NV>(buf + lhalo, g);
          }
        }
        __syncthreads();
        if (row_ok && in_tile) {
          // all D neighbour reads first (one wait), then the multiply-adds
          f2 g[D][NV];
#pragma unroll
          for (int d = 0; d < D; ++d) lds_getv<NV>(buf + lown - d * width, g[d]);     // dy[oy][q - d]
#pragma unroll
          for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int u = 0; u < S; ++u) {
              if (u + S * d < K) {
                const int kx = u + S * d;
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                  const int sl = (tt * S + ky) % K;                     // input row oy S + ky - pad_t
#pragma unroll
                  for (int i = 0; i < NV; ++i) {
                    dacc[sl][u][i] = w[ky * K + kx][i] * g[d][i] + dacc[sl][u][i];
                    wacc[ky * K + kx][i] = xt[sl][u][i] * g[d][i] + wacc[ky * K + kx][i];
                  }
                }
              }
            }
          }
        }
        // ---- input rows oy S + v - pad_t (v < S) are complete
#pragma unroll
        for (int v = 0; v < S; ++v) {
          const int sl = (tt * S + v) % K;
          if (c_rng[v].has(oy)) {                                       // uniform
            const uint32_t gr = (uint32_t)(oy * S + v - a.pad_t) * xrow_b;
#pragma unroll
            for (int u = 0; u < S; ++u) {
              if (cok[u]) {
                f2 g[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) g[i] = ACTM != 0 ? dacc[sl][u][i] * dsw[sl][u][i] : dacc[sl][u][i];
                if (a.epi.beta) {
                  f2 old[NV];
                  unpackv<NV>(ldg<NV>(go_img, xoff[u], gr), old);
#pragma unroll
                  for (int i = 0; i < NV; ++i) g[i] += old[i];
                }
                stg<NV>(go_img, xoff[u], gr, g);
                if (want_stats) {
                  f2 x[NV];
                  unpackv<NV>(xw[sl][u], x);
#pragma unroll
                  for (int i = 0; i < NV; ++i) {
                    st[0][i] += g[i];
                    st[1][i] = g[i] * ((x[i] - mu[i]) * rs[i]) + st[1][i];
                  }
                }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < S; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) dacc[sl][u][i] = zero2;
        }
      }
    }
    __syncthreads();
  }
  if (want_stats) {
    float stf[2][CPT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) { stf[r][2 * i] = st[r][i].x; stf[r][2 * i + 1] = st[r][i].y; }
    block_channel_sums<CPT, 2>(a, l, C, stf, a.epi.stat_partials, red);
  }
  float wf[K * K][CPT];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int i = 0; i < NV; ++i) { wf[t][2 * i] = wacc[t][i].x; wf[t][2 * i + 1] = wacc[t][i].y; }
  block_channel_sums<CPT, K * K>(a, l, C, wf, a.ws, red);
}

// =====================================================================================================
// Forward, round 6: the round-2..5 kernel (k_fwd_lx) rewritten with the instruction economy of k_bwd_one (the r05 counters show the forward
// kernels 29-37 % issue-stalled at four waves per SIMD): raw buffer loads (uniform row offset + a per-thread byte
// offset that is constant over a tile), 2-vectors of adjacent channels, the uniform row conditions as ranges of the
// step index, all K neighbour reads of a step in front of its multiply-adds.  Same march: thread = output column x
// CPT channels over the input rows t = r + pad_t of its tile, ceil(K / S) output rows in flight.
// Every load of a step is issued on every step, at an in-range address (row, column and channel clamped): what lies
// outside the image / the channel range is zeroed where the row is consumed.  A load under a branch turns every wait
// on the FIFO into a wait for ALL loads in flight -- the newest row included, i.e. the whole HBM latency once per row
// step (r03o: the waits of the first kernels were all vmcnt(0)).  The FIFO of rows in flight is a ring of NF register
// sets addressed by static indices (the row loop is unrolled over lcm(U0, NF) steps): no set is ever copied.
template <int K, int S, int CPT, int ACTM>
__global__ __launch_bounds__(THREADS, (K == 3 && CPT == 2) ? 6 : 4) void k_fwd_v2(const Args a) {
  constexpr int NV = CPT / 2;
  constexpr int NSL = (K + S - 1) / S;   // output rows in flight
  constexpr int U0 = S * NSL;            // row steps after which the accumulator slots repeat
  constexpr int PF = U0 == 3 ? 5 : (U0 == 5 ? 4 : (U0 == 4 ? 3 : 5));
  constexpr int NF = PF + 1;
  constexpr int U = U0 * NF / gcd_(U0, NF);
  constexpr int HALO = K - S;
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX * S + HALO;
  float* ring = red + red_floats(a.nch, CPT);
  const bool in_tile = l.px < a.TX;
  const char* IN = reinterpret_cast<const char*>(a.in.data);
  char* OUT = reinterpret_cast<char*>(a.out);
  f2 w[K * K][NV], sc[NV], sh[NV], st[2][NV];
  const f2 zero2 = {0.f, 0.f}, one2 = {1.f, 1.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) { sc[i] = one2; sh[i] = zero2; st[0][i] = st[1][i] = zero2; }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      w[t][i] = zero2;
      if (l.active) w[t][i] = *reinterpret_cast<const f2*>(a.w + (size_t)t * C + l.c + 2 * i);
    }
  if (l.active && a.in.scale) { loadv<NV>(a.in.scale + l.c, sc); loadv<NV>(a.in.shift + l.c, sh); }
  const bool want_stats = a.stat_partials != nullptr;
  const uint32_t cc2 = (uint32_t)(l.c < C ? l.c : 0) * 2u;
  const uint32_t irow_b = (uint32_t)W * a.in.ld * 2, orow_b = (uint32_t)a.ow * a.ldo * 2;   // < 2^31 per image (host check)
  const uint32_t lown = (uint32_t)(l.px * S * width + l.chunk * CPT);          // ring slot of the own window column 0
  const uint32_t lhalo = (uint32_t)((a.TX * S + l.px) * width + l.chunk * CPT);

  for (int tile = l.tile0; tile < l.tend; tile += l.tstep) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty = rr / a.tiles_x, tx = rr - ty * a.tiles_x;
    const int oy0 = ty * a.TY, oy1 = min(a.oh, oy0 + a.TY);
    const int ox = tx * a.TX + l.px;
    const bool xok = l.active && ox < a.ow;
    const int wc0 = tx * a.TX * S - a.pad_l;           // input column of window column 0
    bool mok[S];
    uint32_t coff[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int ix = wc0 + l.px * S + j;
      mok[j] = l.active && ix >= 0 && ix < W;
      coff[j] = (uint32_t)min(max(ix, 0), W - 1) * (uint32_t)(a.in.ld * 2) + cc2;
    }
    const int ixh = wc0 + a.TX * S + l.px;
    const bool hown = HALO > 0 && l.px < HALO && in_tile;
    const bool hok = hown && l.c < C && ixh >= 0 && ixh < W;
    const uint32_t hoff = hown ? (uint32_t)min(max(ixh, 0), W - 1) * (uint32_t)(a.in.ld * 2) + cc2 : coff[0];
    const uint32_t ooff = (uint32_t)min(ox, a.ow - 1) * (uint32_t)(a.ldo * 2) + cc2;
    const brsrc_t in_img = make_rsrc(IN + (size_t)n * H * irow_b);
    const brsrc_t out_img = make_rsrc(OUT + (size_t)n * a.oh * orow_b);
    // steps t = input row + pad_t, from the first row of output row oy0; the input row exists: pad_t <= t < H + pad_t;
    // output row oy completes at t = oy S + K - 1
    const int t0 = oy0 * S, t_last = (oy1 - 1) * S + K - 1;
    URange row_rng, out_rng;
    row_rng.set(max(t0, a.pad_t), min(t_last + 1, H + a.pad_t));
    out_rng.set(t0 + K - 1, t_last + 1);
    f2 acc[NSL][NV];
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2)
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[s2][i] = zero2;
    RawV<NV> fm[NF][S], fh[NF];
    auto load_row = [&](int t, RawV<NV> (&dst)[S], RawV<NV>& hdst) {
      const uint32_t ro = (uint32_t)min(max(t - a.pad_t, 0), H - 1) * irow_b;
#pragma unroll
      for (int j = 0; j < S; ++j) dst[j] = ldg<NV>(in_img, coff[j], ro);
      if (HALO > 0) hdst = ldg<NV>(in_img, hoff, ro);
    };
    auto transform = [&](const RawV<NV>& raw, bool ok, f2 (&v)[NV]) {
      f2 x[NV];
      unpackv<NV>(raw, x);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const f2 z = x[i] * sc[i] + sh[i];
        f2 y;
        if (ACTM == 1) { y.x = z.x * sigmoidf_(z.x); y.y = z.y * sigmoidf_(z.y); }
        else if (ACTM == 2) { y.x = act_other_(a.in.act, z.x); y.y = act_other_(a.in.act, z.y); }
        else y = z;
        v[i] = ok ? y : zero2;                      // 'SAME' padding is zero in the activated domain
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(t0 + i, fm[i], fh[i]);
    for (int tb = t0; tb <= t_last; tb += U) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int t = tb + tt;
        constexpr int dummy_ = 0; (void)dummy_;
        load_row(t + PF, fm[(tt + PF) % NF], fh[(tt + PF) % NF]);
        const bool row_ok = row_rng.has(t);          // uniform
        float* buf = ring + (t & 1) * WIN * width;
        if (row_ok && in_tile) {
#pragma unroll
          for (int j = 0; j < S; ++j) {
            f2 v[NV];
            transform(fm[tt % NF][j], mok[j], v);
            lds_putv<NV>(buf + lown + j * width, v);
          }
          if (hown) {
            f2 v[NV];
            transform(fh[tt % NF], hok, v);
            lds_putv<NV>(buf + lhalo, v);
          }
        }
        __syncthreads();
        if (row_ok && in_tile) {
          f2 x[K][NV];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) lds_getv<NV>(buf + lown + kx * width, x[kx]);
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
              if (((tt - ky) % S + S) % S == 0) {          // static: this input row feeds output row (t - ky) / S
                const int sl = slot_of(fdiv_(tt - ky, S), NSL);
#pragma unroll
                for (int i = 0; i < NV; ++i) acc[sl][i] = w[ky * K + kx][i] * x[kx][i] + acc[sl][i];
              }
            }
          }
        }
        if (((tt - (K - 1)) % S + S) % S == 0) {           // static: output row (t - K + 1) / S is complete
          const int sl = slot_of(fdiv_(tt - (K - 1), S), NSL);
          if (out_rng.has(t) && xok) {
            const uint32_t oro = (uint32_t)((t - (K - 1)) / S) * orow_b;
            stg<NV>(out_img, ooff, oro, acc[sl]);
            if (want_stats) {
#pragma unroll
              for (int i = 0; i < NV; ++i) {
                f2 v;
                v.x = bf2f(f2bf(acc[sl][i].x)); v.y = bf2f(f2bf(acc[sl][i].y));
                st[0][i] += v;
                st[1][i] = v * v + st[1][i];
              }
            }
          }
#pragma unroll
          for (int i = 0; i < NV; ++i) acc[sl][i] = zero2;
        }
      }
    }
    __syncthreads();      // the next tile's first row reuses the ring slot of this tile's last rows
  }
  if (want_stats) {
    float stf[2][CPT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) { stf[r][2 * i] = st[r][i].x; stf[r][2 * i + 1] = st[r][i].y; }
    block_channel_sums<CPT, 2>(a, l, C, stf, a.stat_partials, red);
  }
}

// ------------------------------------------------------------------------------------- host
inline int pick_nch(int nvec, int maxch) {
  if (nvec <= maxch) return nvec;
  int best = maxch;
  double best_u = 0.0;
  for (int n = maxch; n >= maxch / 2; --n) {
    const int groups = (nvec + n - 1) / n;
    const double u = (double)nvec / (groups * n) * ((THREADS / n) * n) / THREADS;
    if (u > best_u + 1e-9) { best_u = u; best = n; }
  }
  return best;
}

// XCD-aware (tile slot, channel group) -> block map (lane_setup): needs a whole number of slots per XCD.
// EDET_DWM_XCD=0|1|2 overrides (lab switch).
inline void xcd_map(Args& a) {
  const char* e = getenv("EDET_DWM_XCD");
  const int mode = (e && e[0]) ? atoi(e) : 2;
  a.xcd = 0;
  if (mode == 0 || a.P < 8 || (mode == 1 && a.ngroups == 1)) return;
  a.P -= a.P % 8;
  a.xcd = mode;
}

// space_w / space_h: extent of the marched tile space (output pixels; for dgrad the q / oy step space)
template <int CPT>
inline void plan(Args& a, int C, int n, int space_w, int space_h, int max_p, int k, int slack_h = 0, int p_small = 2048) {
  const int nvec = (C + CPT - 1) / CPT;
  // <= 128 contiguous bytes per pixel and workgroup.  (r02t lab: 64- or 32-byte channel groups for the two-channel
  // kernels -- wider column tiles, half the column halo -- are slower: backward 11.04 -> 11.40 / 12.01 ms, forward
  // 4.77 -> 4.87 / 5.01 ms over the 15 layer shapes.)
  const char* gb_env = getenv("EDET_DWM_GB");      // lab switch: bytes of one pixel's channel group per workgroup
  const int gb = (gb_env && gb_env[0]) ? atoi(gb_env) : 128;
  a.nch = pick_nch(nvec, gb / (CPT * 2));
  a.ngroups = (nvec + a.nch - 1) / a.nch;
  a.TX = THREADS / a.nch;
  {
    // Balanced row tiles: at most 80 rows on the 160 / 320-row maps, 40 below (fixed 32-row tiles left a short last
    // tile that still pays the K - 1 halo rows and the pipeline fill; taller tiles than this leave the 80-row maps
    // with too few tiles to fill the chip).  r02i / r02j lab, 15 depthwise layer shapes of D0 640x640 batch 128:
    // backward 12.08 -> 11.49 (40) -> 10.81 ms (80), forward 5.08 -> 4.71 -> 4.59 ms.
    // slack_h: rows of the marched space beyond the map (the padding rows of k_bwd_one's shifted row space) that must
    // not cost an extra tile
    const int cap = space_h >= 160 ? 80 : 40;
    const int nt = max(1, (space_h - slack_h + cap - 1) / cap);
    a.TY = (space_h + nt - 1) / nt;
  }
  if (a.TY > space_h) a.TY = space_h;
  a.tiles_x = (space_w + a.TX - 1) / a.TX;
  a.tiles_y = (space_h + a.TY - 1) / a.TY;
  a.ntiles = n * a.tiles_x * a.tiles_y;
  // Persistent workgroups x channel groups.  r03d lab (scripts/kernel_lab.py --ab EDET_DWM_P=4096,2048,8192, the 15
  // depthwise layer shapes of D0 640x640 batch 128): the 3x3 layers on the 160 / 320-row maps want many short-lived
  // workgroups (8192: 320x320x32 fused backward 1.00 -> 0.86 ms, forward 0.61 -> 0.52 ms), every other layer fewer,
  // longer-lived ones (2048: 40x40x480 k5 backward 0.65 -> 0.54, 20x20x1152 k5 0.46 -> 0.38 ms); over the 15 shapes
  // backward 11.21 -> 10.52 ms, forward 4.84 -> 4.67 ms against round 2's 4096 everywhere.  r06l lab, the round-6 kernels:
  // the forward (four waves per SIMD now) wants 4096 on those layers (3.39 -> 3.34 ms over the 15 shapes, 20x20x1152 k3 0.071
  // -> 0.064 ms), the one-pass backward keeps 2048 (7.14 / 7.16 / 7.35 ms at default / 2048 / 4096).  EDET_DWM_P overrides.
  const char* p_env = getenv("EDET_DWM_P");
  int P = ((p_env && p_env[0]) ? atoi(p_env) : ((k == 3 && a.in.h >= 160) ? 8192 : p_small)) / a.ngroups;
  if (P < 64) P = 64;
  if (P > max_p) P = max_p;
  if (P > a.ntiles) P = a.ntiles;
  a.P = P;
  xcd_map(a);
}

}  // namespace dwm

// return 1 = handled, 0 = not applicable (caller falls back), < 0 = error
int dwm_try_fwd(const edet_tview_t* in, const float* weight, int k, int s, void* out, int ldo,
                float* stat_partials, int* nparts_out, hipStream_t st) {
  using namespace dwm;
  if (in->gate || in->c % 8 != 0) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.w = weight; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  // the kernel's 32-bit offsets inside one image
  if ((int64_t)in->h * in->w * in->ld * 2 >= 0x7fffffffLL || (int64_t)a.oh * a.ow * ldo * 2 >= 0x7fffffffLL) return 0;
  // 3x3: four channels per thread on the large maps, two (six rows of loads in flight, more waves) from 40 x 40 OUTPUT
  // pixels down -- r04 lab, D0 640x640 batch 128: 40x40x64 0.0340 (4) / 0.0278 ms (2), 40x40x480 0.162 / 0.140, 80x80x240
  // stride 2 0.166 / 0.142, 20x20x1152 0.097 / 0.089, but 80x80x64 0.057 / 0.068 and 320x320x32 0.43 / 0.56.  By the map,
  // not the batch.  EDET_DWM_FWD_CPT=4|2 overrides (lab switch).
  const char* fc = getenv("EDET_DWM_FWD_CPT");
  const bool c2 = (fc && fc[0]) ? fc[0] == '2' : a.oh * a.ow <= 40 * 40;
  const int actm = in->act == EDET_ACT_NONE ? 0 : (in->act == EDET_ACT_SWISH ? 1 : 2);
#define DWM_FWD2(K_, S_, CPT_)                                                            \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, a.ow, a.oh, EDET_MAX_PARTS, K_, 0, 4096);                 \
    const size_t lds0 = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                  \
    const size_t ring = (size_t)2 * (a.TX * S_ + K_ - S_) * a.nch * CPT_ * sizeof(float); \
    const dim3 grid(a.P * a.ngroups), block(THREADS);                                     \
    if (actm == 0) edet_launch(k_fwd_v2<K_, S_, CPT_, 0>, grid, block, lds0 + ring, st, a);      \
    else if (actm == 1) edet_launch(k_fwd_v2<K_, S_, CPT_, 1>, grid, block, lds0 + ring, st, a); \
    else edet_launch(k_fwd_v2<K_, S_, CPT_, 2>, grid, block, lds0 + ring, st, a);                \
  } while (0)
  if (k == 3 && s == 1 && c2) DWM_FWD2(3, 1, 2);
  else if (k == 3 && s == 2 && c2) DWM_FWD2(3, 2, 2);
  else if (k == 3 && s == 1) DWM_FWD2(3, 1, 4);
  else if (k == 3 && s == 2) DWM_FWD2(3, 2, 4);
  else if (k == 5 && s == 1) DWM_FWD2(5, 1, 2);
  else if (k == 5 && s == 2) DWM_FWD2(5, 2, 2);
  else return 0;
#undef DWM_FWD2
  if (nparts_out) *nparts_out = a.P;
  EDET_LAUNCH_CHECK("edet_dw_fwd(march)");
  return 1;
}

// channels per thread of the 3x3 stride-2 gradient kernels: 4 on the large maps, 2 (deeper load FIFO, more waves) up to
// 80 x 80 input pixels -- r04 lab, D0 640x640 batch 128, data + weight gradient: 320x320x96 2.55 (4) / 2.77 ms (2),
// 80x80x240 0.529 (4) / 0.460 ms (2).  By the map, not the batch (the parity runs launch what the batch-128 step launches).
// EDET_DWM_S2_CPT="<dgrad><wgrad>" (e.g. "24") overrides (lab switch).
static int s2_cpt(int which, int hw) {
  const char* e = getenv("EDET_DWM_S2_CPT");
  if (!e || strlen(e) < 2) return hw > 80 * 80 ? 4 : 2;
  return e[which] == '2' ? 2 : 4;
}

int dwm_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, int k, int s, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st) {
  using namespace dwm;
  if (in->gate || in->c % 8 != 0 || !workspace) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  const bool oact = in->act > EDET_ACT_SWISH;      // relu / relu6 / hswish: the OACT instantiations
  a.in = *in; a.gy = *dy; a.ws = reinterpret_cast<float*>(workspace);
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  const int64_t kkc = (int64_t)k * k * in->c;
  int max_p = (int)((int64_t)(workspace_bytes / sizeof(float)) / kkc);
  if (max_p < 1) return 0;
  if (max_p > 1024) max_p = 1024;
  const bool gbn = dy->a != nullptr;
#define DWM_WG(K_, S_, CPT_)                                                              \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, a.ow, a.oh, max_p, K_);                                   \
    const size_t lds = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                   \
    const size_t ring = (size_t)2 * (a.TX * S_ + K_ - S_) * a.nch * CPT_ * sizeof(float); \
    const dim3 grid(a.P * a.ngroups), block(THREADS);                                     \
    if (gbn) { if (oact) edet_launch(k_wgrad_lx<K_, S_, CPT_, true, true>, grid, block, lds + ring, st, a); else edet_launch(k_wgrad_lx<K_, S_, CPT_, true, false>, grid, block, lds + ring, st, a); }          \
    else { if (oact) edet_launch(k_wgrad_lx<K_, S_, CPT_, false, true>, grid, block, lds + ring, st, a); else edet_launch(k_wgrad_lx<K_, S_, CPT_, false, false>, grid, block, lds + ring, st, a); }             \
  } while (0)
  if (k == 3 && s == 1) DWM_WG(3, 1, 4);
  else if (k == 3 && s == 2 && s2_cpt(1, in->h * in->w) == 2) DWM_WG(3, 2, 2);
  else if (k == 3 && s == 2) DWM_WG(3, 2, 4);
  else if (k == 5 && s == 1) DWM_WG(5, 1, 2);
  else if (k == 5 && s == 2) DWM_WG(5, 2, 2);
  else return 0;
#undef DWM_WG
  EDET_LAUNCH_CHECK("edet_dw_bwd_weight(march)");
  if (edet_reduce_partials(a.ws, a.P, kkc, dweight, st) != 0) return -2;
  return 1;
}

int dwm_try_dgrad(const edet_gview_t* dy, const float* weight, int k, int s, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st) {
  using namespace dwm;
  if (in->gate || epi->dgate || in->c % 8 != 0) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  const bool oact = in->act > EDET_ACT_SWISH;      // relu / relu6 / hswish: the OACT instantiations
  a.in = *in; a.gy = *dy; a.w = weight; a.epi = *epi;
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  const int QW = (in->w + a.pad_l + s - 1) / s, QH = (in->h + a.pad_t + s - 1) / s;
  const bool gbn = dy->a != nullptr;
#define DWM_DG(K_, S_, CPT_)                                                              \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, QW, QH, EDET_MAX_PARTS, K_);                              \
    const size_t lds = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                   \
    const size_t ring = (size_t)2 * (a.TX + (K_ + S_ - 1) / S_ - 1) * a.nch * CPT_ * sizeof(float); \
    const dim3 grid(a.P * a.ngroups), block(THREADS);                                     \
    if (gbn) { if (oact) edet_launch(k_dgrad_lx<K_, S_, CPT_, true, true>, grid, block, lds + ring, st, a); else edet_launch(k_dgrad_lx<K_, S_, CPT_, true, false>, grid, block, lds + ring, st, a); }          \
    else { if (oact) edet_launch(k_dgrad_lx<K_, S_, CPT_, false, true>, grid, block, lds + ring, st, a); else edet_launch(k_dgrad_lx<K_, S_, CPT_, false, false>, grid, block, lds + ring, st, a); }             \
  } while (0)
  if (k == 3 && s == 1) DWM_DG(3, 1, 4);
  else if (k == 3 && s == 2 && s2_cpt(0, in->h * in->w) == 2) DWM_DG(3, 2, 2);
  else if (k == 3 && s == 2) DWM_DG(3, 2, 4);
  else if (k == 5 && s == 1) DWM_DG(5, 1, 2);
  else if (k == 5 && s == 2) DWM_DG(5, 2, 2);
  else return 0;
#undef DWM_DG
  if (nparts_out) *nparts_out = a.P;
  EDET_LAUNCH_CHECK("edet_dw_bwd_data(march)");
  return 1;
}

// data + weight gradient in one pass (k_bwd_one, any stride): 1 = handled, 0 = not applicable (the caller runs the two
// separate kernels)
int dwm_try_bwd_fused(const edet_gview_t* dy, const float* weight, int k, int s, const edet_tview_t* in,
                      const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  using namespace dwm;
  // k = 3, stride 1: 4 channels per thread (8-byte loads, 2 waves/SIMD) wins on the large maps, 2 channels per thread
  // (3-4 waves/SIMD) on the small ones (r02: 320x320x32 1.08 vs 1.55 ms, 40x40x64 1.25 vs 1.05 ms; r04, 64 channels:
  // 80x80 0.147 (2) -> 0.137 ms (4), 40x40 0.0485 -> 0.0480 with the threshold at 80x80, 0.057 with 4 channels there too)
  const char* c4_env = getenv("EDET_DWM_C4_MINHW");      // lab switch
  const bool k3c4 = (int64_t)in->h * in->w >= (c4_env && c4_env[0] ? atoi(c4_env) : 80 * 80);
  const char* one_env = getenv("EDET_DWM_ONE");          // lab switch: 0 = the two-kernel path (k_wgrad_lx, k_dgrad_lx)
  if (one_env && one_env[0] == '0') return 0;
  if ((s != 1 && s != 2) || (k != 3 && k != 5) || in->gate || epi->dgate || in->c % 8 != 0 || !workspace) return 0;
  if ((int64_t)in->h * in->w * in->ld * 2 >= 0x7fffffffLL) return 0;      // the kernel's 32-bit offsets inside one image
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.gy = *dy; a.w = weight; a.epi = *epi; a.ws = reinterpret_cast<float*>(workspace);
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  const int64_t kkc = (int64_t)k * k * in->c;
  int max_p = (int)((int64_t)(workspace_bytes / sizeof(float)) / kkc);
  if (max_p < 1) return 0;
  if (max_p > EDET_MAX_PARTS) max_p = EDET_MAX_PARTS;
  const bool gbn = dy->a != nullptr;
  {
    const int QW = (in->w + a.pad_l + s - 1) / s, QH = (in->h + a.pad_t + s - 1) / s;
    const int actm = in->act == EDET_ACT_NONE ? 0 : (in->act == EDET_ACT_SWISH ? 1 : 2);
#define DWM_ONE(K_, S_, CPT_)                                                             \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, QW, QH, max_p, K_, QH - a.oh);                            \
    const size_t lds = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                   \
    const size_t ring = (size_t)2 * (a.TX + (K_ + S_ - 1) / S_ - 1) * a.nch * CPT_ * sizeof(float); \
    const dim3 grid(a.P * a.ngroups), block(THREADS);                                     \
    if (gbn) {                                                                            \
      if (actm == 0) edet_launch(k_bwd_one<K_, S_, CPT_, true, 0>, grid, block, lds + ring, st, a);      \
      else if (actm == 1) edet_launch(k_bwd_one<K_, S_, CPT_, true, 1>, grid, block, lds + ring, st, a); \
      else edet_launch(k_bwd_one<K_, S_, CPT_, true, 2>, grid, block, lds + ring, st, a);                \
    } else {                                                                              \
      if (actm == 0) edet_launch(k_bwd_one<K_, S_, CPT_, false, 0>, grid, block, lds + ring, st, a);      \
      else if (actm == 1) edet_launch(k_bwd_one<K_, S_, CPT_, false, 1>, grid, block, lds + ring, st, a); \
      else edet_launch(k_bwd_one<K_, S_, CPT_, false, 2>, grid, block, lds + ring, st, a);                \
    }                                                                                     \
  } while (0)
    if (k == 3 && s == 1 && k3c4) DWM_ONE(3, 1, 4);
    else if (k == 3 && s == 1) DWM_ONE(3, 1, 2);
    else if (k == 3 && s == 2) DWM_ONE(3, 2, 2);
    else if (k == 5 && s == 1) DWM_ONE(5, 1, 2);
    else DWM_ONE(5, 2, 2);
#undef DWM_ONE
    if (nparts_out) *nparts_out = a.P;
    EDET_LAUNCH_CHECK("edet_dw_bwd(one pass)");
    if (edet_reduce_partials(a.ws, a.P, kkc, dweight, st) != 0) return -2;
  }
  return 1;
}
