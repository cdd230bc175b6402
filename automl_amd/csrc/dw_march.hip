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
// through act'(z), the optional accumulate and the BatchNorm backward sums in the row epilogue).
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
};

struct Lane {
  int chunk, px, g, p, c;
  bool active;
};
template <int CPT>
__device__ __forceinline__ Lane lane_setup(const Args& a, int C) {
  Lane l;
  l.chunk = threadIdx.x % a.nch;
  l.px = threadIdx.x / a.nch;
  l.g = blockIdx.x % a.ngroups;
  l.p = blockIdx.x / a.ngroups;
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

template <int K, int S, int CPT, bool OACT>
__global__ __launch_bounds__(THREADS) void k_fwd_lx(const Args a) {
  constexpr int NSL = (K + S - 1) / S;   // output rows in flight
  constexpr int U0 = S * NSL;            // row steps after which the window slots repeat
  // The FIFO of rows in flight is a ring of NF register sets addressed by static indices: no set is ever copied into
  // its neighbour (r03q: the shift fm[i] = fm[i+1] at every step survived unrolling as NF moves per array and U0 steps,
  // about as many VALU instructions as the 3x3 multiply-adds).  The row loop is unrolled over lcm(U0, NF) steps, so NF
  // is chosen as a divisor or a small multiple of U0 near the depth the lab found best (6 rows at stride 1, 4 at 2).
  constexpr int PF = U0 == 3 ? 5 : (U0 == 5 ? 4 : (U0 == 4 ? 3 : (U0 == 6 ? 5 : PfDepth<S, CPT>::fwd)));
  constexpr int NF = PF + 1;
  constexpr int U = U0 * NF / gcd_(U0, NF);   // static unroll of the row loop
  constexpr int HALO = K - S;            // window columns beyond TX * S
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX * S + HALO;
  float* ring = red + red_floats(a.nch, CPT);          // [2][WIN][width]
  const bool in_tile = l.px < a.TX;                    // thread owns window columns
  const bf16_t* IN = reinterpret_cast<const bf16_t*>(a.in.data);
  float w[K * K][CPT], sc[CPT], sh[CPT];
  float st[2][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) { sc[e] = 1.f; sh[e] = 0.f; st[0][e] = st[1][e] = 0.f; }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < CPT; ++e) w[t][e] = l.active ? a.w[(size_t)t * C + l.c + e] : 0.f;
  if (l.active && a.in.scale) { loadf<CPT>(a.in.scale + l.c, sc); loadf<CPT>(a.in.shift + l.c, sh); }
  const bool want_stats = a.stat_partials != nullptr;

  for (int tile = l.p; tile < a.ntiles; tile += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty = rr / a.tiles_x, tx = rr - ty * a.tiles_x;
    const int oy0 = ty * a.TY, oy1 = min(a.oh, oy0 + a.TY);
    const int ox = tx * a.TX + l.px;
    const bool xok = l.active && ox < a.ow;
    const int wc0 = tx * a.TX * S - a.pad_l;           // input column of window column 0
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
    bf16_t* obase = a.out + ((size_t)n * a.oh * a.ow + ox) * a.ldo + l.c;
    float acc[NSL][CPT];
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int e = 0; e < CPT; ++e) acc[s][e] = 0.f;

    const int t0 = (oy0 * S / U0) * U0, t_last = (oy1 - 1) * S + K - 1;   // t = input row + pad_t
    Raw<CPT> fm[NF][S], fh[NF];                             // ring: step tt consumes set tt % NF, fills (tt + PF) % NF
#pragma unroll
    for (int i = 0; i <= PF; ++i) {
      fh[i] = raw_zero<CPT>();
#pragma unroll
      for (int j = 0; j < S; ++j) fm[i][j] = raw_zero<CPT>();
    }
    // Every load of a step is issued on every step, at an in-range address (row, column and channel clamped): what
    // lies outside the image / the channel range is zeroed where the row is consumed (mok / hok below).  A load under
    // a branch turns every wait on the FIFO into a wait for ALL loads in flight -- the newest row included, i.e. the
    // whole HBM latency once per row step (r03o: the waits of these kernels were all vmcnt(0)).
    int64_t coff[S], hoff;
#pragma unroll
    for (int j = 0; j < S; ++j) coff[j] = (int64_t)min(max(wc0 + l.px * S + j, 0), W - 1) * a.in.ld;
    hoff = hown ? (int64_t)min(max(ixh, 0), W - 1) * a.in.ld : coff[0];
    const bf16_t* ibase_c = IN + ((int64_t)n * H * W * a.in.ld + (l.c < C ? l.c : 0));
    auto load_row = [&](int t, Raw<CPT> (&dst)[S], Raw<CPT>& hdst) {
      const bf16_t* rp = ibase_c + (int64_t)min(max(t - a.pad_t, 0), H - 1) * W * a.in.ld;
#pragma unroll
      for (int j = 0; j < S; ++j) dst[j] = raw_load<CPT>(rp + coff[j]);
      if (HALO > 0) hdst = raw_load<CPT>(rp + hoff);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(t0 + i, fm[i], fh[i]);
    for (int tb = t0; tb <= t_last; tb += U) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int t = tb + tt;
        load_row(t + PF, fm[(tt + PF) % NF], fh[(tt + PF) % NF]);      // PF input rows in flight per thread
        const int r = t - a.pad_t;
        const bool row_ok = t <= t_last && r >= 0 && r < H;   // uniform over the workgroup
        float* buf = ring + (t & 1) * WIN * width;
        if (row_ok && in_tile) {
#pragma unroll
          for (int j = 0; j < S; ++j) {
            float x[CPT];
            raw_unpack<CPT>(fm[tt % NF][j], x);
            view_act<CPT, OACT>(a.in, sc, sh, x);
            if (!mok[j]) {
#pragma unroll
              for (int e = 0; e < CPT; ++e) x[e] = 0.f;   // 'SAME' padding is zero in the activated domain
            }
            lds_put<CPT>(buf + (l.px * S + j) * width + l.chunk * CPT, x);
          }
          if (hown) {
            float x[CPT];
            raw_unpack<CPT>(fh[tt % NF], x);
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
              if ((tt - ky) % S == 0) {                    // static: this input row feeds output (t - ky) / S
                const int sl = slot_of((tt - ky) / S, NSL);
#pragma unroll
                for (int e = 0; e < CPT; ++e) acc[sl][e] = fmaf(w[ky * K + kx][e], x[e], acc[sl][e]);
              }
            }
          }
        }
        if ((tt - (K - 1)) % S == 0) {                     // static: output row (t - K + 1) / S is complete
          const int sl = slot_of((tt - (K - 1)) / S, NSL);
          const int oy = (t - (K - 1)) / S;
          if (t <= t_last && t >= K - 1 && oy >= oy0 && oy < oy1 && xok) {
            store_bf<CPT>(obase + (size_t)oy * a.ow * a.ldo, acc[sl]);
            if (want_stats) {
#pragma unroll
              for (int e = 0; e < CPT; ++e) {
                const float v = bf2f(f2bf(acc[sl][e]));
                st[0][e] += v;
                st[1][e] = fmaf(v, v, st[1][e]);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < CPT; ++e) acc[sl][e] = 0.f;
        }
      }
    }
    __syncthreads();      // the next tile's first row reuses the ring slot of this tile's last rows
  }
  if (want_stats) block_channel_sums<CPT, 2>(a, l, C, st, a.stat_partials, red);
}

// weight gradient with the activated input row exchanged through LDS (see k_fwd_lx); dy is the thread's own
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

  for (int tile = l.p; tile < a.ntiles; tile += a.P) {
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
  for (int tile = l.p; tile < a.ntiles; tile += a.P) {
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
// Fused backward for stride 1: data gradient AND weight gradient in one march (edet_dw_bwd).
// Both need the same operands -- dy = BN-backward(dz, y) around a pixel and the saved input x of the pixel:
//   d in[iy][ix]   = sum_{ky,kx} w[ky][kx] * dy[iy+p-ky][ix+p-kx]          (then * act'(z), + old, BN sums)
//   dW[ky][kx]    += act(z)[iy][ix]        * dy[iy+p-ky][ix+p-kx]
// so the separate kernels read (dz, y, x) twice and evaluate the BatchNorm backward and the sigmoid of z twice.
// Here thread (column ix, CPT channels) marches over the dy rows oy: the transformed dy row goes through the
// LDS ring (K column neighbours come back after the barrier), the thread's own column of x is kept as a
// K-row register window of act(z) (for dW) and act'(z) (for the epilogue of the row that completes), and
// every neighbour value g = dy[oy][ix+p-kx] feeds both accumulations:
//   dacc[ky][.] += w[ky][kx] * g   (input row iy = oy - p + ky)      wacc[ky][kx] += act(z)[oy - p + ky] * g
// One pass: reads dz, y, x once, writes d in once; one sigmoid and one BN-backward FMA pair per element.
// A tile owns the input rows [r0, r1) x its columns: dy rows r0-p .. r1-1+p are marched (the 2p extra rows
// are re-read by the neighbouring tile), x rows outside [r0, r1) enter the window as zeros so that every
// (x pixel, dy pixel) pair is counted by exactly one tile.
template <int K, int CPT, bool GBN, bool OACT>
__global__ __launch_bounds__(THREADS, CPT * K >= 10 ? 2 : 3) void k_bwd_fused(const Args a) {
  // rows of global loads in flight: 3 rows left the march latency-bound (a row step is ~0.2 us of work, HBM latency
  // under load ~2 us); r02i lab: 6 rows (4 with four channels per thread: register budget) 12.08 -> 11.44 ms over
  // the 15 depthwise layer shapes of D0 640x640 batch 128; 8 rows: 11.07 -> 10.92 ms (r02j), not worth the registers
  // ring of NF register sets with static indices (see k_fwd_lx): K = 3 -> 6 sets (5 rows ahead; 3 sets = 2 rows ahead
  // with four channels per thread: register budget of the two-wave kernel), K = 5 -> 5 sets; the row loop is unrolled
  // over lcm(K, NF) steps
  constexpr int PF = K == 3 ? (CPT == 4 ? 2 : 5) : (K == 5 ? 4 : (CPT == 4 ? 4 : 6));
  constexpr int NF = PF + 1;
  constexpr int U = K * NF / gcd_(K, NF);
  constexpr int PD = (K - 1) / 2;             // 'SAME' padding of an odd kernel at stride 1
  extern __shared__ float red[];
  const int C = a.in.c, H = a.in.h, W = a.in.w;
  const Lane l = lane_setup<CPT>(a, C);
  const int width = a.nch * CPT;
  const int WIN = a.TX + K - 1;
  float* ring = red + red_floats(a.nch, CPT);
  const bool in_tile = l.px < a.TX;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.in.data);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gy.dz);
  const bf16_t* YY = reinterpret_cast<const bf16_t*>(a.gy.y);
  bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);
  float w[K * K][CPT], wacc[K * K][CPT], sc[CPT], sh[CPT], ga[CPT], gb[CPT], gc[CPT], mu[CPT], rs[CPT];
  float st[2][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) {
    sc[e] = 1.f; sh[e] = 0.f; ga[e] = 1.f; gb[e] = 0.f; gc[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f;
    st[0][e] = st[1][e] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      w[t][e] = l.active ? a.w[(size_t)t * C + l.c + e] : 0.f;
      wacc[t][e] = 0.f;
    }
  const bool want_stats = a.epi.stat_partials != nullptr;
  constexpr bool other = OACT;
  const bool swish = !OACT && a.in.act == EDET_ACT_SWISH;
  if (l.active) {
    if (a.in.scale) { loadf<CPT>(a.in.scale + l.c, sc); loadf<CPT>(a.in.shift + l.c, sh); }
    if (GBN) { loadf<CPT>(a.gy.a + l.c, ga); loadf<CPT>(a.gy.b + l.c, gb); loadf<CPT>(a.gy.cc + l.c, gc); }
    if (want_stats) { loadf<CPT>(a.epi.mean + l.c, mu); loadf<CPT>(a.epi.rstd + l.c, rs); }
  }

  for (int tile = l.p; tile < a.ntiles; tile += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty_ = rr / a.tiles_x, tx_ = rr - ty_ * a.tiles_x;
    const int q0 = tx_ * a.TX;
    const int q = q0 + l.px;                                   // own column (input == dy column at stride 1)
    const bool qok = l.active && in_tile && q < W;
    const int r0 = ty_ * a.TY, r1 = min(H, r0 + a.TY);
    // halo columns of the dy window: px < PD -> left column q0 - PD + px (wc = px), PD <= px < 2 PD -> right
    // column q0 + TX + (px - PD) (wc = TX + px)
    const bool hown = l.px < K - 1 && in_tile;
    const int qh = l.px < PD ? q0 - PD + l.px : q0 + a.TX + (l.px - PD);
    const int wch = l.px < PD ? l.px : a.TX + l.px;
    const bool hok = hown && l.c < C && qh >= 0 && qh < W;
    const size_t img = (size_t)n * H * W;
    float dacc[K][CPT], xt[K][CPT], dsw[K][CPT];
    Raw<CPT> xw[K];                                            // raw x of the window rows (BN backward sums)
#pragma unroll
    for (int s = 0; s < K; ++s) {
      xw[s] = raw_zero<CPT>();
#pragma unroll
      for (int e = 0; e < CPT; ++e) { dacc[s][e] = 0.f; xt[s][e] = 0.f; dsw[s][e] = 1.f; }
    }
    // step t: dy row oy = r0 - PD + t, newest x row iy = r0 + t, completed row iy = r0 - 2 PD + t
    const int t_last = (r1 - r0) - 1 + 2 * PD;
    Raw<CPT> fz[NF], fy[GBN ? NF : 1], fhz[NF], fhy[GBN ? NF : 1], fx[NF];     // step tt consumes set tt % NF
#pragma unroll
    for (int i = 0; i <= PF; ++i) {
      fz[i] = fhz[i] = fx[i] = raw_zero<CPT>();
      if (GBN) fy[i] = fhy[i] = raw_zero<CPT>();
    }
    if (!GBN) fy[0] = fhy[0] = raw_zero<CPT>();
    // every load of a step is issued on every step at an in-range address (see k_fwd_lx); rows / columns / channels
    // outside the tile or the image are zeroed where they are consumed (row_ok, qok, hok, xin below)
    const int cc = l.c < C ? l.c : 0;
    const int64_t qoff = (int64_t)min(q, W - 1), qhoff = hown ? (int64_t)min(max(qh, 0), W - 1) : qoff;
    auto load_step = [&](int t, Raw<CPT>& z, Raw<CPT>& y, Raw<CPT>& hz, Raw<CPT>& hy, Raw<CPT>& x) {
      const int oyc = min(max(r0 - PD + t, 0), H - 1);
      const size_t rowoff = (img + (size_t)oyc * W) * a.gy.ld + cc;
      z = raw_load<CPT>(DZ + rowoff + (size_t)qoff * a.gy.ld);
      if (GBN) y = raw_load<CPT>(YY + rowoff + (size_t)qoff * a.gy.ld);
      hz = raw_load<CPT>(DZ + rowoff + (size_t)qhoff * a.gy.ld);
      if (GBN) hy = raw_load<CPT>(YY + rowoff + (size_t)qhoff * a.gy.ld);
      x = raw_load<CPT>(X + (img + (size_t)min(r0 + t, H - 1) * W + qoff) * a.in.ld + cc);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_step(i, fz[i], fy[GBN ? i : 0], fhz[i], fhy[GBN ? i : 0], fx[i]);
    for (int tb = 0; tb <= t_last; tb += U) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int t = tb + tt;
        constexpr int dummy0 = 0; (void)dummy0;
        const int fc = tt % NF, fn = (tt + PF) % NF;       // set consumed / filled at this step
        load_step(t + PF, fz[fn], fy[GBN ? fn : 0], fhz[fn], fhy[GBN ? fn : 0], fx[fn]);
        const int oy = r0 - PD + t;
        // ---- newest x row enters the window at slot (tt + K - 1) % K
        {
          constexpr int dummy = 0; (void)dummy;
          const int sl = slot_of(tt + K - 1, K);
          const int iy = r0 + t;
          float x[CPT], z[CPT];
          raw_unpack<CPT>(fx[fc], x);
          xw[sl] = fx[fc];
          const bool xin = iy < r1 && qok;
#pragma unroll
          for (int e = 0; e < CPT; ++e) z[e] = fmaf(x[e], sc[e], sh[e]);
          if (swish) {
#pragma unroll
            for (int e = 0; e < CPT; ++e) {
              const float sg = sigmoidf_(z[e]);
              xt[sl][e] = xin ? z[e] * sg : 0.f;
              dsw[sl][e] = sg * (1.f + z[e] * (1.f - sg));
            }
          } else if (other) {
#pragma unroll
            for (int e = 0; e < CPT; ++e) {
              xt[sl][e] = xin ? act_other_(a.in.act, z[e]) : 0.f;
              dsw[sl][e] = act_other_grad_(a.in.act, z[e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < CPT; ++e) { xt[sl][e] = xin ? z[e] : 0.f; dsw[sl][e] = 1.f; }
          }
        }
        // ---- dy row oy through the LDS ring
        const bool row_ok = t <= t_last && oy >= 0 && oy < H;   // uniform
        float* buf = ring + (t & 1) * WIN * width;
        if (row_ok && in_tile) {
          float g[CPT];
          raw_unpack<CPT>(fz[fc], g);
          if (GBN) {
            float y[CPT];
            raw_unpack<CPT>(fy[GBN ? fc : 0], y);
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[e] = qok ? fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e])) : 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[e] = qok ? g[e] : 0.f;
          }
          lds_put<CPT>(buf + (l.px + PD) * width + l.chunk * CPT, g);
          if (hown) {
            raw_unpack<CPT>(fhz[fc], g);
            if (GBN) {
              float y[CPT];
              raw_unpack<CPT>(fhy[GBN ? fc : 0], y);
#pragma unroll
              for (int e = 0; e < CPT; ++e) g[e] = hok ? fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e])) : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < CPT; ++e) g[e] = hok ? g[e] : 0.f;
            }
            lds_put<CPT>(buf + wch * width + l.chunk * CPT, g);
          }
        }
        __syncthreads();
        if (row_ok && in_tile) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            float g[CPT];
            lds_get<CPT>(buf + (l.px + 2 * PD - kx) * width + l.chunk * CPT, g);   // dy[oy][q + PD - kx]
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
              const int sl = slot_of(tt + ky, K);                                 // input row oy - PD + ky
#pragma unroll
              for (int e = 0; e < CPT; ++e) {
                dacc[sl][e] = fmaf(w[ky * K + kx][e], g[e], dacc[sl][e]);
                wacc[ky * K + kx][e] = fmaf(xt[sl][e], g[e], wacc[ky * K + kx][e]);
              }
            }
          }
        }
        // ---- input row iy = oy - PD is complete (slot tt % K)
        {
          const int sl = slot_of(tt, K);
          const int iy = oy - PD;
          if (t <= t_last && iy >= r0 && iy < r1 && qok) {
            const size_t off = (img + (size_t)iy * W + q) * a.in.ld + l.c;
            float g[CPT], x[CPT];
#pragma unroll
            for (int e = 0; e < CPT; ++e) g[e] = dacc[sl][e] * dsw[sl][e];
            if (a.epi.beta) {
              float old[CPT];
              raw_unpack<CPT>(raw_load<CPT>(GO + off), old);
#pragma unroll
              for (int e = 0; e < CPT; ++e) g[e] += old[e];
            }
            store_bf<CPT>(GO + off, g);
            if (want_stats) {
              raw_unpack<CPT>(xw[sl], x);
#pragma unroll
              for (int e = 0; e < CPT; ++e) {
                st[0][e] += g[e];
                st[1][e] = fmaf(g[e], (x[e] - mu[e]) * rs[e], st[1][e]);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < CPT; ++e) dacc[sl][e] = 0.f;
        }
      }
    }
    __syncthreads();
  }
  if (want_stats) block_channel_sums<CPT, 2>(a, l, C, st, a.epi.stat_partials, red);
  block_channel_sums<CPT, K * K>(a, l, C, wacc, a.ws, red);
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

// space_w / space_h: extent of the marched tile space (output pixels; for dgrad the q / oy step space)
template <int CPT>
inline void plan(Args& a, int C, int n, int space_w, int space_h, int max_p, int k) {
  const int nvec = (C + CPT - 1) / CPT;
  // <= 128 contiguous bytes per pixel and workgroup.  (r02t lab: 64- or 32-byte channel groups for the two-channel
  // kernels -- wider column tiles, half the column halo -- are slower: backward 11.04 -> 11.40 / 12.01 ms, forward
  // 4.77 -> 4.87 / 5.01 ms over the 15 layer shapes.)
  a.nch = pick_nch(nvec, 128 / (CPT * 2));
  a.ngroups = (nvec + a.nch - 1) / a.nch;
  a.TX = THREADS / a.nch;
  {
    // Balanced row tiles: at most 80 rows on the 160 / 320-row maps, 40 below (fixed 32-row tiles left a short last
    // tile that still pays the K - 1 halo rows and the pipeline fill; taller tiles than this leave the 80-row maps
    // with too few tiles to fill the chip).  r02i / r02j lab, 15 depthwise layer shapes of D0 640x640 batch 128:
    // backward 12.08 -> 11.49 (40) -> 10.81 ms (80), forward 5.08 -> 4.71 -> 4.59 ms.
    const int cap = space_h >= 160 ? 80 : 40;
    const int nt = (space_h + cap - 1) / cap;
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
  // backward 11.21 -> 10.52 ms, forward 4.84 -> 4.67 ms against round 2's 4096 everywhere.  EDET_DWM_P overrides.
  const char* p_env = getenv("EDET_DWM_P");
  int P = ((p_env && p_env[0]) ? atoi(p_env) : ((k == 3 && a.in.h >= 160) ? 8192 : 2048)) / a.ngroups;
  if (P < 64) P = 64;
  if (P > max_p) P = max_p;
  if (P > a.ntiles) P = a.ntiles;
  a.P = P;
}

// lab switch EDET_DWM_ROUNDS=r: persistent workgroups = r x what the chip holds of THIS kernel at once (occupancy query)
inline void replan_rounds(Args& a, const void* fn, size_t lds, int max_p) {
  const char* e = getenv("EDET_DWM_ROUNDS");
  if (!e || !e[0]) return;
  const int slots = edet_resident_wgs(fn, THREADS, lds);
  if (slots <= 0) return;
  int P = (int)(atof(e) * slots) / a.ngroups;
  if (P < 1) P = 1;
  if (P > max_p) P = max_p;
  if (P > a.ntiles) P = a.ntiles;
  a.P = P;
}

}  // namespace dwm

// return 1 = handled, 0 = not applicable (caller falls back), < 0 = error
int dwm_try_fwd(const edet_tview_t* in, const float* weight, int k, int s, void* out, int ldo,
                float* stat_partials, int* nparts_out, hipStream_t st) {
  using namespace dwm;
  if (in->gate || in->c % 8 != 0) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  const bool oact = in->act > EDET_ACT_SWISH;      // relu / relu6 / hswish: the OACT instantiations
  a.in = *in; a.w = weight; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
#define DWM_FWD(K_, S_, CPT_)                                                             \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, a.ow, a.oh, EDET_MAX_PARTS, K_);                          \
    const size_t lds0 = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                  \
    const size_t ring = (size_t)2 * (a.TX * S_ + K_ - S_) * a.nch * CPT_ * sizeof(float); \
    replan_rounds(a, reinterpret_cast<const void*>(&k_fwd_lx<K_, S_, CPT_, false>), lds0 + ring, EDET_MAX_PARTS); \
    if (oact) edet_launch(k_fwd_lx<K_, S_, CPT_, true>, dim3(a.P * a.ngroups), dim3(THREADS), lds0 + ring, st, a); \
    else edet_launch(k_fwd_lx<K_, S_, CPT_, false>, dim3(a.P * a.ngroups), dim3(THREADS), lds0 + ring, st, a); \
  } while (0)
  // 3x3: four channels per thread on the large maps, two (six rows of loads in flight, more waves) from 40 x 40 OUTPUT
  // pixels down -- r04 lab, D0 640x640 batch 128: 40x40x64 0.0340 (4) / 0.0278 ms (2), 40x40x480 0.162 / 0.140, 80x80x240
  // stride 2 0.166 / 0.142, 20x20x1152 0.097 / 0.089, but 80x80x64 0.057 / 0.068 and 320x320x32 0.43 / 0.56.  By the map,
  // not the batch.  EDET_DWM_FWD_CPT=4|2 overrides (lab switch).
  const char* fc = getenv("EDET_DWM_FWD_CPT");
  const bool c2 = (fc && fc[0]) ? fc[0] == '2' : a.oh * a.ow <= 40 * 40;
  if (k == 3 && s == 1 && c2) DWM_FWD(3, 1, 2);
  else if (k == 3 && s == 2 && c2) DWM_FWD(3, 2, 2);
  else if (k == 3 && s == 1) DWM_FWD(3, 1, 4);
  else if (k == 3 && s == 2) DWM_FWD(3, 2, 4);
  else if (k == 5 && s == 1) DWM_FWD(5, 1, 2);
  else if (k == 5 && s == 2) DWM_FWD(5, 2, 2);
  else return 0;
#undef DWM_FWD
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

// fused data + weight gradient (stride 1): 1 = handled, 0 = not applicable (the caller runs the two kernels)
int dwm_try_bwd_fused(const edet_gview_t* dy, const float* weight, int k, int s, const edet_tview_t* in,
                      const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  using namespace dwm;
  // k = 3: 4 channels per thread (8-byte loads, 2 waves/SIMD) wins on the large maps, 2 channels per thread
  // (3-4 waves/SIMD) on the small ones (measured r02: 320x320x32 1.08 vs 1.55 ms, 40x40x64 1.25 vs 1.05 ms)
  const char* c4_env = getenv("EDET_DWM_C4_MINHW");      // lab switch
  // r04 lab (D0 640x640 batch 128, 64 channels): 80x80 0.147 (2 channels) -> 0.137 ms (4), 40x40 0.0485 -> 0.0480 with the
  // threshold at 80x80, 0.057 with 4 channels there too
  const bool k3c4 = (int64_t)in->h * in->w >= (c4_env && c4_env[0] ? atoi(c4_env) : 80 * 80);
  if (s != 1 || (k != 3 && k != 5) || in->gate || epi->dgate || in->c % 8 != 0 || !workspace) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  const bool oact = in->act > EDET_ACT_SWISH;      // relu / relu6 / hswish: the OACT instantiations
  a.in = *in; a.gy = *dy; a.w = weight; a.epi = *epi; a.ws = reinterpret_cast<float*>(workspace);
  a.oh = in->h; a.ow = in->w;
  a.pad_t = a.pad_l = (k - 1) / 2;
  const int64_t kkc = (int64_t)k * k * in->c;
  int max_p = (int)((int64_t)(workspace_bytes / sizeof(float)) / kkc);
  if (max_p < 1) return 0;
  if (max_p > EDET_MAX_PARTS) max_p = EDET_MAX_PARTS;
  const bool gbn = dy->a != nullptr;
#define DWM_FUSED(K_, CPT_)                                                               \
  do {                                                                                    \
    plan<CPT_>(a, in->c, in->n, in->w, in->h, max_p, K_);                                 \
    const size_t lds = (size_t)red_floats(a.nch, CPT_) * sizeof(float);                   \
    const size_t ring = (size_t)2 * (a.TX + K_ - 1) * a.nch * CPT_ * sizeof(float);       \
    replan_rounds(a, reinterpret_cast<const void*>(&k_bwd_fused<K_, CPT_, true, false>), lds + ring, max_p); \
    const dim3 grid(a.P * a.ngroups), block(THREADS);                                     \
    if (gbn) { if (oact) edet_launch(k_bwd_fused<K_, CPT_, true, true>, grid, block, lds + ring, st, a); else edet_launch(k_bwd_fused<K_, CPT_, true, false>, grid, block, lds + ring, st, a); }             \
    else { if (oact) edet_launch(k_bwd_fused<K_, CPT_, false, true>, grid, block, lds + ring, st, a); else edet_launch(k_bwd_fused<K_, CPT_, false, false>, grid, block, lds + ring, st, a); }                \
  } while (0)
  if (k == 3 && k3c4) DWM_FUSED(3, 4);
  else if (k == 3) DWM_FUSED(3, 2);
  else DWM_FUSED(5, 2);
#undef DWM_FUSED
  if (nparts_out) *nparts_out = a.P;
  EDET_LAUNCH_CHECK("edet_dw_bwd(fused)");
  if (edet_reduce_partials(a.ws, a.P, kkc, dweight, st) != 0) return -2;
  return 1;
}
