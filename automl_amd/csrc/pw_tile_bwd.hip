// One-pass TILED backward of a pointwise (1x1) convolution on gfx950: data gradient AND weight gradient from a single
// read of (dz, y, x).
//
// The wave-private one-pass kernel of pw_stream.hip (k_pw_bwd_fused) keeps the whole dW block of a wave in registers,
// which limits it to the expand / project layers of the large maps (dW <= ~45 tiles of 16 x 16) and makes it own every
// register of a compute unit.  The layers it cannot take -- the 58 64 -> 64 layers of the BiFPN and the class / box
// towers, the 40x40 / 20x20 backbone projections (K up to 1152 channels against N <= 128) -- ran the two tiled kernels
// of pw_big.hip back to back, each of which streams (dz, y, x): 7 tensor streams per layer where one pass needs 4.
//
// Here a 256-thread workgroup owns a slice of KT input channels and a contiguous range of 64-row steps.  Per step
//   * every thread loads whole 16-byte chunks of x (its KT-channel slice), dz and y (all N <= NT output channels) one
//     step ahead, applies the producer's BatchNorm + activation + SE gate to x and the BatchNorm backward
//     dy = a*dz + b*y + c to the gradient, and parks both ROW-major in LDS as bf16 (tiles Xt [64][KT], Dt [64][NT]);
//   * data gradient  C[64][KT] = Dt . Wl^T on v_mfma_f32_32x32x16_bf16, both operands plain 16-byte LDS reads (the weight
//     slice Wl [KT][NT] is resident in LDS for the whole kernel); C goes through an fp32 LDS tile to the thread that
//     loaded the same (row, chunk) of x, which chains act'(z) / the accumulate / the BatchNorm-backward sums / the SE
//     gate sums and stores 16 bytes;
//   * weight gradient dW[KT][NT] += Xt^T . Dt with the contraction over the step's 64 rows: both fragments come through
//     the gfx950 LDS transpose read (ds_read_b64_tr_b16), the accumulators (KT*NT/256 registers per lane) live across
//     the whole row range.
// Two barriers per step, no atomics: per-channel sums are kept in registers across the steps and combined through LDS
// in a fixed order; SE gate sums are flushed per image into per-workgroup slots that k_gate_finish adds in slot order;
// the dW partial of every row split goes to the workspace and edet_reduce_partials sums the splits in order.  Layers
// with K > KT run one workgroup column per slice (each re-reads the N <= 128 wide gradient, from the XCD's L2 when the
// slices of a split run side by side: block b -> XCD b % 8).
//
// Reference call sites replaced: TF Conv2DBackpropInput + Conv2DBackpropFilter of the 1x1 convolutions of
// efficientdet/backbone/efficientnet_model.py:304-312,345-353 and efficientdet/tf2/efficientdet_keras.py:195-207,286-290,
// 459-464,546-556 under the GradientTape of efficientdet/tf2/train_lib.py:623-669.
#include <stdlib.h>

#include "common.h"

namespace pwt {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v_t;

constexpr int THREADS = 256;
constexpr int RS = 64;          // rows per step

struct Args {
  edet_gview_t gv;      // dy: N = gv.c channels
  edet_tview_t tv;      // conv input view: K = tv.c channels
  const bf16_t* W;      // [K][ldw], n contiguous
  int ldw;
  edet_bwd_epi_t epi;
  float* ws;            // [S][K][N] fp32 partial weight gradients
  float* gate_ws;       // [GP][images][K] SE gate-gradient slots (gated input only)
  unsigned char* dump;  // THREADS * 16 bytes: where the stores of rows / channels outside the tensor go
  int M, K, N;
  int hwp;              // rows of one "image" of the step grid: pixels per image when the input is gated, else M
  int spi;              // steps per image = ceil(hwp / 64)
  int T;                // steps in all
  int sps;              // steps per split
  int S;                // row splits (= statistic partial rows)
  int nsl;              // KT-channel slices of K
};

__device__ __forceinline__ void unpack8(const uint4 raw, float x[8]) {
  x[0] = __uint_as_float(raw.x << 16); x[1] = __uint_as_float(raw.x & 0xffff0000u);
  x[2] = __uint_as_float(raw.y << 16); x[3] = __uint_as_float(raw.y & 0xffff0000u);
  x[4] = __uint_as_float(raw.z << 16); x[5] = __uint_as_float(raw.z & 0xffff0000u);
  x[6] = __uint_as_float(raw.w << 16); x[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float x[8]) {
  uint4 o;
  o.x = pack2bf(x[0], x[1]); o.y = pack2bf(x[2], x[3]);
  o.z = pack2bf(x[4], x[5]); o.w = pack2bf(x[6], x[7]);
  return o;
}
// the first `nvalid` (<= 8) bf16 elements of a chunk, the others zeroed (padding columns may hold anything)
__device__ __forceinline__ uint4 keep_first(uint4 v, int nvalid) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (2 * i >= nvalid) w[i] = 0u;
    else if (2 * i + 1 >= nvalid) w[i] &= 0xffffu;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// MFMA fragment "8 consecutive rows (row0 ..) of one channel" of a row-major bf16 tile through the LDS transpose read:
// in a 16-lane group lane i addresses 4 consecutive channels (8 bytes) of row row0 + i / 4 -- together a [4 rows][16
// channels] block starting at channel col0 -- and receives the 4 rows of channel col0 + i (pw_stream.hip, r03m).
__device__ __forceinline__ bf16x8 column_frag_tr(const unsigned char* tile, int stride, int row0, int col0, int fi) {
  const unsigned char* p = tile + (row0 + (fi >> 2)) * stride + (col0 + (fi & 3) * 4) * 2;
  typedef __attribute__((address_space(3))) bf16x4v_t* lds_ptr_t;
  const bf16x4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p));
  const bf16x4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p + 4 * stride));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int lds_bytes(int KT, int NT, int NSL = 1, bool coefs = true) {
  return RS * (KT * 2 + 16) + RS * (NT * 2 + 16) + KT * (NSL * NT * 2 + 16) + RS * (KT * 4 + 16) +
         (coefs ? (2 * KT + 3 * NSL * NT) * 4 : 0);
}
constexpr int min_blocks(int KT, int NT, bool xgen, int NSL = 1) {
  return (lds_bytes(KT, NT, NSL) > 80 * 1024 || NSL > 3) ? 1 : ((xgen || NSL > 1 || lds_bytes(KT, NT, NSL) > 53 * 1024) ? 2 : 3);
}

// XM = 0: the input is a plain stored tensor (no BatchNorm / activation / gate on load, no statistic or gate sums in the
// epilogue) -- the pointwise half of every SeparableConv2D of the BiFPN and the towers.  XM = 1: the general view.
// XM = 2: the SE-gated view of an MBConv projection with the gate-gradient sums in the epilogue (dgate[n][k] = sum_hw
// d * act(z)): the data gradient is stored as it is and act(z) * gate is ALREADY in the LDS operand tile, so the epilogue
// adds d * x~ from its own slot and the flush divides by the gate -- no second sigmoid per element, no raw x kept in
// registers (the product is the bf16-rounded operand: 2^-9 relative noise per term of a sum over the image).
// NSL > 1 (r04): N up to NSL * 128 output channels in NSL column slices of the gradient per 64-row step -- the weight
// slice [KT][N] stays resident, the data-gradient accumulators run over the slices of a step, the weight-gradient
// accumulators (NSL sets) over the whole row range, one gradient slice is in flight while the previous one is on the
// matrix cores.  The 20x20 projections (1152 -> 192 / 320), the 40 -> 240 expansion and the 810-column class-predict
// layers (K = 64: 7 slices, 224 accumulator registers, one workgroup per compute unit).
template <int KT, int NT, bool GBN, int XM, bool OACT, int NSL = 1, int PF = 1>
__global__ __launch_bounds__(THREADS, min_blocks(KT, NT, XM != 0, NSL)) void k_pw_bwd_tile(const Args a) {
  static_assert(PF == 1 || PF == 2, "one or two steps of loads in flight");
  static_assert(NSL == 1 || NT == 128, "column slices are 128 channels wide");
  constexpr bool XGEN = XM != 0;      // the input view carries BatchNorm / activation / gate, or the epilogue sums
  constexpr bool XGATE = XM == 2;     // SE-gated input with gate-gradient sums: the epilogue needs no raw x
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int SX = KT * 2 + 16, SD = NT * 2 + 16, SW = NSL * NT * 2 + 16, SC = KT * 4 + 16;
  constexpr int NTT = NSL * NT;      // output channels covered
  unsigned char* Xt = smem;
  unsigned char* Dt = Xt + RS * SX;
  unsigned char* Wl = Dt + RS * SD;
  unsigned char* Ct = Wl + KT * SW;
  // per-channel coefficients (scale, shift | a, b, c): read from LDS where they are used, so that they do not occupy 40
  // registers across the matrix phase
  float* cfx = reinterpret_cast<float*>(Ct + RS * SC);     // [2][KT]
  float* cfd = cfx + 2 * KT;                               // [3][NTT]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  // block -> (split, slice): the slices of one split back to back on one XCD (block b runs on XCD b % 8)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int slice = q % a.nsl;
  const int split = (q / a.nsl) * 8 + xcd;
  if (split >= a.S) return;
  const int k0 = slice * KT;
  const int t0 = split * a.sps, t1 = min(a.T, t0 + a.sps);

  // staging / epilogue geometry: a thread owns one 8-channel chunk column of each operand and NP rows of a step
  constexpr int CPRX = KT / 8, RPPX = THREADS / CPRX, NPX = RS / RPPX;
  constexpr int CPRD = NT / 8, RPPD = THREADS / CPRD, NPD = RS / RPPD;
  const int xc = tid % CPRX, xr0 = tid / CPRX;
  const int dc = tid % CPRD, dr0 = tid / CPRD;
  const int kx = k0 + xc * 8;
  const bool x_ok = kx < a.K;
  const int kxc = x_ok ? kx : 0;            // column actually addressed (loads are unconditional)
  const int nd = dc * 8;                     // channel of the gradient chunk within its column slice
  // valid elements of the gradient chunk of slice j as a bit mask (all ones: the whole chunk; zero: a chunk past N) --
  // the padding columns of dy may hold anything
  auto slice_mask = [&](int j) {
    const int n0 = j * NT + nd;
    return n0 < a.N ? keep_first(make_uint4(~0u, ~0u, ~0u, ~0u), a.N - n0) : make_uint4(0, 0, 0, 0);
  };
  uint4 dmask[NSL <= 2 ? NSL : 1];           // (more slices: recomputed per slice, a handful of scalar-ish instructions)
  if constexpr (NSL <= 2) {
#pragma unroll
    for (int j = 0; j < NSL; ++j) dmask[j] = slice_mask(j);
  }

  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.tv.data);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gv.dz);
  const bf16_t* DY = reinterpret_cast<const bf16_t*>(a.gv.y);
  bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);
  const int ldx = a.tv.ld, ldd = a.gv.ld;

  const bool affine = XGEN && a.tv.scale != nullptr;
  const bool swish = XGEN && !OACT && a.tv.act == EDET_ACT_SWISH;
  constexpr bool other = XGEN && OACT;
  const bool gated = XGEN && a.tv.gate != nullptr;
  const bool want_stats = XGEN && a.epi.stat_partials != nullptr;
  const bool want_gate = XGEN && a.epi.dgate != nullptr;
  const bool beta = a.epi.beta != 0;

  // ---- prologue: the weight slice -> LDS (rows past K and columns past N zero), per-channel coefficients -> registers
  constexpr int CPRW = NTT / 8;              // chunks per weight row
  for (int idx = tid; idx < KT * CPRW; idx += THREADS) {
    const int k = idx / CPRW, c = idx - k * CPRW;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k0 + k < a.K && c * 8 < a.N) {
      v = *reinterpret_cast<const uint4*>(a.W + (size_t)(k0 + k) * a.ldw + c * 8);
      if (a.N - c * 8 < 8) v = keep_first(v, a.N - c * 8);
    }
    *reinterpret_cast<uint4*>(Wl + k * SW + c * 16) = v;
  }
  if constexpr (XGEN) {
    for (int k = tid; k < KT; k += THREADS) {
      const bool ok = affine && k0 + k < a.K;
      cfx[k] = ok ? a.tv.scale[k0 + k] : 1.f;
      cfx[KT + k] = ok ? a.tv.shift[k0 + k] : 0.f;
    }
  }
  if constexpr (GBN) {
    for (int n = tid; n < NTT; n += THREADS) {
      const bool ok = n < a.N;
      cfd[n] = ok ? a.gv.a[n] : 0.f;
      cfd[NTT + n] = ok ? a.gv.b[n] : 0.f;
      cfd[2 * NTT + n] = ok ? a.gv.cc[n] : 0.f;
    }
  }

  // ---- accumulators
  constexpr int DT = KT / 64;                 // data-gradient column tiles per wave (row tile = wave & 1)
  constexpr int WKT = KT / 64, WNT = NT / 64; // weight-gradient tiles per wave: WKT x WNT
  f32x16 accw[NSL][WKT][WNT];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int i = 0; i < WKT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[sl][i][j][e] = 0.f;
  float s1[XGEN ? 8 : 1], s2[XGEN ? 8 : 1];      // BatchNorm-backward sums, or (s1) the SE gate sums: never both
  if constexpr (XGEN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; }
  }

  // ---- loads in flight (one step ahead)
  // a register set of loads in flight: PF = 1 one set (step t+1 requested while step t is worked on), PF = 2 two sets used
  // alternately (step t+2 requested at step t: twice the bytes in flight per workgroup)
  struct LoadSet {
    uint4 x[NPX], z[NPD], y[GBN ? NPD : 1];
    float g[XGEN ? 8 : 1];
  };
  LoadSet LA, LB;
  auto geometry = [&](int t, int& img, int& r0, int& nvalid) {
    img = t / a.spi;
    const int qs = t - img * a.spi;
    r0 = img * a.hwp + qs * RS;
    nvalid = min(RS, a.hwp - qs * RS);
  };
  auto issue_x = [&](int t, LoadSet& L) {
    int img, r0, nvalid;
    geometry(t, img, r0, nvalid);
    // one uniform 64-bit base per tensor and step + a 32-bit byte offset per lane
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(X + (size_t)r0 * ldx);
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int r = min(xr0 + RPPX * i, nvalid - 1);         // rows past the step re-read its last row
      L.x[i] = *reinterpret_cast<const uint4*>(xb + (uint32_t)((r * ldx + kxc) * 2));
    }
    if constexpr (XGEN) {
      if (gated) loadf8(a.tv.gate + (size_t)img * a.K + kxc, L.g);
    }
  };
  auto issue_d = [&](int t, int j, LoadSet& L) {      // column slice j of the gradient rows of step t
    int img, r0, nvalid;
    geometry(t, img, r0, nvalid);
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(DZ + (size_t)r0 * ldd);
    const unsigned char* yb = reinterpret_cast<const unsigned char*>(DY + (size_t)r0 * ldd);
    const int n0 = j * NT + nd;
    const int ndc = n0 < a.N ? n0 : 0;                       // chunks past N re-read chunk 0 (masked where they are used)
#pragma unroll
    for (int i = 0; i < NPD; ++i) {
      const int r = min(dr0 + RPPD * i, nvalid - 1);
      const uint32_t off = (uint32_t)((r * ldd + ndc) * 2);
      L.z[i] = *reinterpret_cast<const uint4*>(zb + off);
      if constexpr (GBN) L.y[i] = *reinterpret_cast<const uint4*>(yb + off);
    }
  };

  const int r_lane = lane & 31, h_lane = lane >> 5;
  const int fi = lane & 15, fg = (lane >> 4) & 1;
  const int rt = wave & 1;                    // data gradient: row tile of this wave
  const int wkt0 = (wave & 1) * WKT, wnt0 = (wave >> 1) * WNT;

  if (t0 < t1) {
    issue_x(t0, LA); issue_d(t0, 0, LA);
    if constexpr (PF == 2) { issue_x(min(t0 + 1, t1 - 1), LB); issue_d(min(t0 + 1, t1 - 1), 0, LB); }
  }
  __syncthreads();                            // Wl complete
  // one step: the loads of step t are in L; the same set then receives the loads of step tn
  auto step = [&](int t, int tn, LoadSet& L) {
    int img, r0, nvalid;
    geometry(t, img, r0, nvalid);

    // ---- stage step t: transformed operands -> LDS
    uint4 xcur[(XGEN && !XGATE) ? NPX : 1];
    float gt[XGEN ? 8 : 1];
    float sc[XGEN ? 8 : 1], sh[XGEN ? 8 : 1];
    if constexpr (XGEN) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gt[e] = gated ? L.g[e] : 1.f;
      loadf8(cfx + xc * 8, sc);
      loadf8(cfx + KT + xc * 8, sh);
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int r = xr0 + RPPX * i;
      uint4 v = L.x[i];
      if constexpr (XGEN) {
        if constexpr (!XGATE) xcur[i] = L.x[i];
        if (affine || swish || other || gated) {
          float x[8];
          unpack8(v, x);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(x[e], sc[e], sh[e]);
            x[e] = (other ? act_other_(a.tv.act, z) : (swish ? swishf_(z) : z)) * gt[e];
          }
          v = pack8(x);
        }
      }
      if (!(x_ok && r < nvalid)) v = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(Xt + r * SX + xc * 16) = v;
    }
    f32x16 accd[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) accd[i][e] = 0.f;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      // ---- stage column slice sl of dy
      {
        float ga[GBN ? 8 : 1], gb[GBN ? 8 : 1], gc[GBN ? 8 : 1];
        if constexpr (GBN) {
          loadf8(cfd + sl * NT + nd, ga); loadf8(cfd + NTT + sl * NT + nd, gb); loadf8(cfd + 2 * NTT + sl * NT + nd, gc);
        }
        uint4 smask;
        if constexpr (NSL <= 2) smask = dmask[sl]; else smask = slice_mask(sl);
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
          const int r = dr0 + RPPD * i;
          uint4 v = L.z[i];
          if constexpr (GBN) {
            float g[8], y[8];
            unpack8(L.z[i], g);
            unpack8(L.y[i], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e]));
            v = pack8(g);
          }
          const uint4 mk = r < nvalid ? smask : make_uint4(0, 0, 0, 0);
          v = make_uint4(v.x & mk.x, v.y & mk.y, v.z & mk.z, v.w & mk.w);
          *reinterpret_cast<uint4*>(Dt + r * SD + dc * 16) = v;
        }
      }
      // ---- request what comes next: the next slice of this step, or step t+1 (the last step re-requests itself: the
      // loads stay unconditional)
      if (sl + 1 < NSL) {
        issue_d(t, sl + 1, L);
      } else {
        issue_x(min(tn, t1 - 1), L);
        issue_d(min(tn, t1 - 1), 0, L);
      }
      __syncthreads();

      // ---- data gradient: C[row][k] += sum_n Dt[row][n] * Wl[k][sl*NT + n]  (D[i = k][j = row])
#pragma unroll
      for (int kk = 0; kk < NT / 16; ++kk) {
        const int koff = (kk * 16 + h_lane * 8) * 2;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Dt + (rt * 32 + r_lane) * SD + koff);
#pragma unroll
        for (int i = 0; i < DT; ++i) {
          const int ct = (wave >> 1) * DT + i;
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wl + (ct * 32 + r_lane) * SW + sl * NT * 2 + koff);
          accd[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, accd[i], 0, 0, 0);
        }
      }
      // ---- weight gradient: dW[k][sl*NT + n] += sum_rows Xt[row][k] * Dt[row][n]  (D[i = k][j = n])
#pragma unroll
      for (int ks = 0; ks < RS / 16; ++ks) {
        const int row0 = ks * 16 + h_lane * 8;
        bf16x8 xf[WKT], df[WNT];
#pragma unroll
        for (int i = 0; i < WKT; ++i) xf[i] = column_frag_tr(Xt, SX, row0, (wkt0 + i) * 32 + fg * 16, fi);
#pragma unroll
        for (int j = 0; j < WNT; ++j) df[j] = column_frag_tr(Dt, SD, row0, (wnt0 + j) * 32 + fg * 16, fi);
#pragma unroll
        for (int i = 0; i < WKT; ++i)
#pragma unroll
          for (int j = 0; j < WNT; ++j)
            accw[sl][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i], df[j], accw[sl][i][j], 0, 0, 0);
      }
      if (sl + 1 < NSL) __syncthreads();       // every wave is done with this slice of Dt
    }
    // C tile: lane holds row rt*32 + r_lane, channels ct*32 + 8g + 4h .. +3
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int ct = (wave >> 1) * DT + i;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Ct + (rt * 32 + r_lane) * SC + (ct * 32 + 8 * g + 4 * h_lane) * 4) =
            make_float4(accd[i][4 * g + 0], accd[i][4 * g + 1], accd[i][4 * g + 2], accd[i][4 * g + 3]);
    }
    // the previous contents of gout (accumulate): requested once the accumulators of the data gradient are dead, in
    // flight across the barrier
    uint4 old[NPX];
    if (beta) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int r = min(xr0 + RPPX * i, nvalid - 1);
        old[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(GO + (size_t)r0 * ldx) +
                                                 (uint32_t)((r * ldx + kxc) * 2));
      }
    }
    __syncthreads();

    // ---- epilogue: the thread that loaded (row, chunk) of x finishes the same chunk of dx
    if constexpr (XGEN && !XGATE) {
      loadf8(cfx + xc * 8, sc);
      loadf8(cfx + KT + xc * 8, sh);
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int r = xr0 + RPPX * i;
      const bool valid = x_ok && r < nvalid;
      const float4 d0 = *reinterpret_cast<const float4*>(Ct + r * SC + xc * 32);
      const float4 d1 = *reinterpret_cast<const float4*>(Ct + r * SC + xc * 32 + 16);
      const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      float g[8];
      float x[XGEN ? 8 : 1];
      if constexpr (XGATE) {
        unpack8(*reinterpret_cast<const uint4*>(Xt + r * SX + xc * 16), x);      // act(z) * gate, this thread's own slot
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = fmaf(d[e], x[e], s1[e]); g[e] = d[e]; }
      } else if constexpr (XGEN) {
        unpack8(xcur[i], x);
        if (want_gate) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = fmaf(x[e], sc[e], sh[e]);
            s1[e] = fmaf(d[e], other ? act_other_(a.tv.act, z) : (swish ? swishf_(z) : z), s1[e]);
            g[e] = d[e];
          }
        } else if (swish) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = d[e] * swish_gradf_(fmaf(x[e], sc[e], sh[e]));
        } else if (other) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = d[e] * act_other_grad_(a.tv.act, fmaf(x[e], sc[e], sh[e]));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = d[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = d[e];
      }
      if (beta) {
        float o[8];
        unpack8(valid ? old[i] : make_uint4(0, 0, 0, 0), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += o[e];
      }
      // rows / channels outside the tensor carry g = 0 (zero operand rows, zero weight rows); their store goes to the dump
      unsigned char* dst = valid ? reinterpret_cast<unsigned char*>(GO + (size_t)r0 * ldx) + (uint32_t)((r * ldx + kx) * 2)
                                 : a.dump + tid * 16;
      *reinterpret_cast<uint4*>(dst) = pack8(g);
      if constexpr (XGEN && !XGATE) {
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += g[e]; s2[e] = fmaf(g[e], x[e], s2[e]); }
        }
      }
    }
    // ---- SE gate sums of an image: flushed when the image (or the range) ends, into this workgroup's slot
    if constexpr (XGEN) {
      if (want_gate && (t + 1 == t1 || (t + 1) / a.spi != img)) {
        __syncthreads();                       // every epilogue read of Ct is done
        float* red = reinterpret_cast<float*>(Ct);      // [RPPX][KT]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[xr0 * KT + xc * 8 + e] = s1[e]; s1[e] = 0.f; }
        __syncthreads();
        if (tid < KT && k0 + tid < a.K) {
          float tot = 0.f;
#pragma unroll 8
          for (int rr = 0; rr < RPPX; ++rr) tot += red[rr * KT + tid];
          const int slot = split - (img * a.spi) / a.sps;
          if constexpr (XGATE) {      // the sums were taken against act(z) * gate
            const float gv = a.tv.gate[(size_t)img * a.K + k0 + tid];
            tot = gv != 0.f ? tot / gv : 0.f;
          }
          a.gate_ws[((size_t)slot * a.tv.n + img) * a.K + k0 + tid] = tot;
        }
        // (the next write of Ct comes after the next step's first barrier)
      }
    }
  };
  if constexpr (PF == 1) {
    for (int t = t0; t < t1; ++t) step(t, t + 1, LA);
  } else {
    for (int t = t0; t < t1; t += 2) {
      step(t, t + 2, LA);
      if (t + 1 < t1) step(t + 1, t + 3, LB);
    }
  }

  // ---- BatchNorm-backward sums of this workgroup's rows: one partial row per split, columns of this slice
  if constexpr (XGEN && !XGATE) {
    if (want_stats) {
      __syncthreads();
      float* red = reinterpret_cast<float*>(Ct);        // [2][RPPX][KT]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[xr0 * KT + xc * 8 + e] = s1[e];
        red[(RPPX + xr0) * KT + xc * 8 + e] = s2[e];
      }
      __syncthreads();
      if (tid < KT && k0 + tid < a.K) {
        float sg = 0.f, sgx = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < RPPX; ++rr) { sg += red[rr * KT + tid]; sgx += red[(RPPX + rr) * KT + tid]; }
        float* dst = a.epi.stat_partials + (size_t)split * 2 * a.K;
        dst[k0 + tid] = sg;
        dst[a.K + k0 + tid] = a.epi.rstd[k0 + tid] * (sgx - a.epi.mean[k0 + tid] * sg);     // sum g*(x-mean)*rstd
      }
    }
  }

  // ---- dW partial of this split: lane holds n = sl*NT + nt*32 + r_lane, k = kt*32 + (e&3) + 8*(e>>2) + 4*h
  float* dst = a.ws + (size_t)split * a.K * a.N;
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int i = 0; i < WKT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = sl * NT + (wnt0 + j) * 32 + r_lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = k0 + (wkt0 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h_lane;
          if (k < a.K && n < a.N) dst[(size_t)k * a.N + n] = accw[sl][i][j][e];
        }
      }
}

// dgate[img][k] += the slots of the workgroups that touched the image, in slot order
__global__ __launch_bounds__(256) void k_gate_finish(const float* __restrict__ gate_ws, int nimg, int K, int spi, int sps,
                                                     float* __restrict__ dgate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= nimg * K) return;
  const int img = idx / K;
  const int first = (img * spi) / sps, last = ((img + 1) * spi - 1) / sps;
  float tot = 0.f;
  for (int p = 0; p <= last - first; ++p) tot += gate_ws[(size_t)p * nimg * K + idx];
  dgate[idx] += tot;
}

inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

template <int KT, int NT, bool GBN, int XM, bool OACT, int NSL = 1, int PF = 1>
int launch(Args& a, int* nparts_out, size_t workspace_bytes, hipStream_t st) {
  auto kern = k_pw_bwd_tile<KT, NT, GBN, XM, OACT, NSL, PF>;
  constexpr size_t lds = lds_bytes(KT, NT, NSL, GBN || XM != 0);
  static const bool lds_ok = lds <= 64 * 1024 ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!lds_ok) return 0;
  static const int resident = edet_resident_wgs(reinterpret_cast<const void*>(kern), THREADS, lds);
  a.nsl = (a.K + KT - 1) / KT;
  // Row splits: one round of resident workgroups (EDET_PWT_ROUNDS rounds), at least EDET_PWT_MINSTEPS steps each,
  // bounded by the statistic partial rows and the workspace.  Every split writes a K x N fp32 partial that
  // edet_reduce_partials reads back.
  // Rounds of resident workgroups.  One round everywhere except where K is cut into MANY slices (r06 lab, D0 640x640 batch
  // 128, EDET_PWT_ROUNDS = 1 / 2 / 3 / 4): 40x40x672->112 (11 slices) 0.355 / 0.299 / 0.259 / 0.274 ms, 20x20x672->192 0.179 /
  // 0.164 / 0.144 / 0.163, 20x20x1152->192 (18 slices, two column slices) 0.284 / 0.210 / 0.261 / 0.230, 20x20x1152->320
  // (18 slices, three column slices, one workgroup per CU) 0.608 / 0.496 / 0.454 / 0.390; every layer with <= 8 slices
  // loses 3-30 % beyond one round.  The slices of a row split re-read the same (dz, y) rows and only share them through
  // the L2 while they walk in step: nothing synchronises them, and the longer the split the further they drift apart.
  int rounds = env_int("EDET_PWT_ROUNDS", 0);
  // (efficientdet-d7x, 384 -> 384 BiFPN / tower layers: 6 slices, three column slices: 96x96 0.239 / 0.214 / 0.162 / 0.183 ms,
  // 48x48 0.094 / 0.077 / 0.076 / 0.076)
  if (rounds <= 0) rounds = NSL >= 3 ? (a.nsl >= 16 ? 4 : (a.nsl >= 4 ? 3 : 1)) : (a.nsl >= 16 ? 2 : (a.nsl >= 10 ? 3 : 1));
  const int slots = (resident > 0 ? resident : 512) * rounds;
  int S = env_int("EDET_PWT_SPLITS", slots / a.nsl);
  const int minsteps = env_int("EDET_PWT_MINSTEPS", 4);
  if (S > a.T / minsteps) S = a.T / minsteps;
  if (S > EDET_MAX_PARTS) S = EDET_MAX_PARTS;
  if (S < 1) S = 1;
  const bool gated = a.tv.gate != nullptr && a.epi.dgate != nullptr;
  for (;;) {
    a.sps = (a.T + S - 1) / S;
    a.S = (a.T + a.sps - 1) / a.sps;
    const int gp = gated ? (a.spi + a.sps - 1) / a.sps + 1 : 0;
    const size_t need = ((size_t)a.S * a.K * a.N + (size_t)gp * a.tv.n * a.K) * sizeof(float) + THREADS * 16;
    if (need <= workspace_bytes) break;
    if (S == 1) return 0;
    S = S / 2;
  }
  a.gate_ws = a.ws + (size_t)a.S * a.K * a.N;
  const int gp = gated ? (a.spi + a.sps - 1) / a.sps + 1 : 0;
  a.dump = reinterpret_cast<unsigned char*>(a.gate_ws + (size_t)gp * a.tv.n * a.K);
  if (nparts_out) *nparts_out = a.S;
  const int grid = (a.S + 7) / 8 * 8 * a.nsl;
  edet_launch(kern, dim3(grid), dim3(THREADS), lds, st, a);
  return 1;
}

}  // namespace pwt

// return 1 = handled, 0 = shape outside the envelope (the caller runs the two-kernel path), < 0 = error
int pwt_try_bwd(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in, const edet_bwd_epi_t* epi,
                int* nparts_out, float* dweight, void* workspace, size_t workspace_bytes, hipStream_t st) {
  using namespace pwt;
  const int N = dy->c, K = in->c;
  if (!workspace || K % 8 != 0 || in->ld % 8 != 0 || dy->ld % 8 != 0 || ldw % 8 != 0 || N > 896 || N < 1) return 0;
  // N % 8 != 0 (the 36-column box-predict layers): only without a BatchNorm backward on dy (its per-channel vectors
  // are read in chunks of 8); the straddling chunk of dy and of the weights is masked
  if (N % 8 != 0 && (dy->a || dy->ld < (N + 7) / 8 * 8 || ldw < (N + 7) / 8 * 8)) return 0;
  if (epi->stat_partials && epi->dgate) return 0;
  if ((size_t)(uintptr_t)workspace % 16 != 0) return 0;
  Args a;
  memset(&a, 0, sizeof(a));
  a.gv = *dy; a.tv = *in; a.W = reinterpret_cast<const bf16_t*>(w); a.ldw = ldw; a.epi = *epi;
  a.ws = reinterpret_cast<float*>(workspace);
  a.M = in->n * in->h * in->w; a.K = K; a.N = N;
  const bool gated = in->gate != nullptr;
  a.hwp = gated ? in->h * in->w : a.M;       // gated input: a step never straddles images (one gate row per step)
  a.spi = (a.hwp + RS - 1) / RS;
  a.T = (gated ? in->n : 1) * a.spi;
  if (a.M < 1) return 0;
  const bool gbn = dy->a != nullptr;
  const bool xgen = in->scale || in->act != EDET_ACT_NONE || in->gate || epi->stat_partials || epi->dgate;
  const bool oact = in->act > EDET_ACT_SWISH;
  // Slice width (r04b lab, D0 640x640 batch 128, KT = 64 against 128): with N > 64 the 128 x 128 tile holds one workgroup
  // per compute unit (103 KB of LDS) and loses -- 672->112 0.435 against 0.652 ms, 480->112 0.225 / 0.240, 480->80 0.217 /
  // 0.232, 240->80 0.119 / 0.129 --; with N <= 64 the wide slice wins on the large maps (80x80x240->40 0.296 against
  // 0.357 ms) and is a wash on the small ones (20x20x320->64 0.0287 / 0.0270): wide from 80 x 80 pixels per image up.
  // EDET_PWT_KT overrides (lab switch).
  const int nt = N <= 64 ? 64 : 128;
  // (decided by the map, not by the batch: the 2-image parity runs then launch the instantiations of the batch-128 step)
  const int kt = K <= 64 ? 64 : env_int("EDET_PWT_KT", (nt == 64 && in->h * in->w >= 3200) ? 128 : 64);
  int rc = 0;
  const bool xgate = in->gate && epi->dgate && !epi->stat_partials && env_int("EDET_PWT_XGATE", 1);
  // (PF = 2, two steps of loads in flight, is not instantiated: r04n lab, +-1 % on every layer shape -- these kernels
  // are not short of bytes in flight)
#define PWT_GO(KT_, NT_, GBN_, XM_, OACT_) rc = launch<KT_, NT_, GBN_, XM_, OACT_>(a, nparts_out, workspace_bytes, st)
#define PWT_X(KT_, NT_, GBN_)                                     \
  do {                                                            \
    if (!xgen) PWT_GO(KT_, NT_, GBN_, 0, false);                  \
    else if (xgate && !oact) PWT_GO(KT_, NT_, GBN_, 2, false);    \
    else if (xgate) PWT_GO(KT_, NT_, GBN_, 2, true);              \
    else if (!oact) PWT_GO(KT_, NT_, GBN_, 1, false);             \
    else PWT_GO(KT_, NT_, GBN_, 1, true);                         \
  } while (0)
#define PWT_G(KT_, NT_)                        \
  do {                                         \
    if (gbn) PWT_X(KT_, NT_, true);            \
    else PWT_X(KT_, NT_, false);               \
  } while (0)
  if (N > 128) {
    // column-sliced instantiations (KT = 64, slices of 128): plain or SE-gated input, swish / linear; 7 slices only for
    // the class-predict layers (no BatchNorm behind them).  EDET_PWT_NSL=0 switches them off (lab switch).
    const int nsl = (N + 127) / 128;
    if (oact || (xgen && !xgate) || !env_int("EDET_PWT_NSL", 1)) return 0;
    // Every K slice re-reads the whole N-wide gradient pair: with more than two slices that only pays on the small maps,
    // where the re-reads come out of the L2 / MALL (r04k, efficientdet-d7x 1536x1536 batch 8: 384->384 at 192x192 10.4 ->
    // 14.7 ms over 18 layers, at 96x96 4.0 -> 5.8, at 24x24 0.92 -> 0.78; D0 20x20x1152->192 0.417 -> 0.284 ms): up to 8192
    // pixels per image
    // (by pixels per image, not rows: the 1-image parity runs then take the same path as the batch-8 step)
    const int hw = in->h * in->w;
    // r06: with 2-4 rounds of row splits (launch) the sliced kernel also wins on efficientdet-d7x's 96 x 96 projections --
    // 1344->224 0.92 -> 0.32 ms, 960->160 0.69 -> 0.23 ms per call at batch 8 -- which the round-4 limit of 8192 pixels
    // left to the two-kernel path; and on its 192 x 192 BiFPN layers (384 -> 384: 0.64 -> 0.48 ms)
    if (K > 128 && hw > env_int("EDET_PWT_NSL_MAXHW", 65536)) return 0;
    const bool wide_expand = nsl > 3 && nsl <= 6 && gbn && !xgen && K <= 128 && env_int("EDET_PWT_WIDE", 1);
    // 7 slices (class predict): up to two K slices (efficientdet-d0 .. d2: 64 / 88 / 112 filters) -- every further slice
    // re-reads the 810-column gradient --, and any K on the small maps, where the two-kernel path would run the generic
    // weight gradient (fp32 atomics) and the re-reads cost nothing (maps up to 8192 pixels per image)
    if (nsl > 3 && !wide_expand && (nsl > 7 || gbn || xgen || (K > 128 && hw > env_int("EDET_PWT_NSL_MAXHW", 65536)))) return 0;
    // r04d lab (D0 640x640 batch 128): 80x80x40->240 0.437 -> 0.210 ms, 40x40x40->240 0.109 -> 0.055, 20x20x1152->192 0.417 ->
    // 0.284, 20x20x672->192 0.266 -> 0.180; three slices hold one workgroup per compute unit (94 KB of LDS) and LOSE on
    // the gated 20x20x1152->320 (0.571 -> 0.612 ms) -- kept all the same: the two-kernel path adds the SE gate-gradient
    // sums with global atomics, this one in a fixed order (EDET_PWT_NSL3=0 switches it off)
    if (nsl == 3 && xgate && !env_int("EDET_PWT_NSL3", 1)) return 0;
#define PWT_N(NSL_, GBN_, XM_) rc = launch<64, 128, GBN_, XM_, false, NSL_>(a, nparts_out, workspace_bytes, st)
#define PWT_NX(NSL_)                                   \
  do {                                                 \
    if (gbn && xgate) PWT_N(NSL_, true, 2);            \
    else if (gbn) PWT_N(NSL_, true, 0);                \
    else if (xgate) PWT_N(NSL_, false, 2);             \
    else PWT_N(NSL_, false, 0);                        \
  } while (0)
    if (nsl == 2) PWT_NX(2);
    else if (nsl == 3) PWT_NX(3);
    else if (wide_expand && nsl == 4) PWT_N(4, true, 0);
    else if (wide_expand) PWT_N(6, true, 0);
    else PWT_N(7, false, 0);
#undef PWT_NX
#undef PWT_N
  } else if (kt == 64 && nt == 64) PWT_G(64, 64);
  else if (kt == 64) PWT_G(64, 128);
  else if (nt == 64) PWT_G(128, 64);
  else PWT_G(128, 128);
#undef PWT_G
#undef PWT_X
#undef PWT_GO
  if (rc <= 0) return rc;
  EDET_LAUNCH_CHECK("edet_pw_bwd(tile)");
  // INVARIANT (deferred reductions): only [a.ws, a.ws + S*K*N) -- the dW partial rows -- outlives this call; the engine
  // hands the NEXT call the workspace behind that range (edet_reduce_deferred_end).  gate_ws / the dump scratch lie
  // beyond it and are consumed by k_gate_finish, launched right here on the same stream: nothing else may read them later.
  if (edet_reduce_partials(a.ws, a.S, (int64_t)K * N, dweight, st) != 0) return -2;
  if (gated && epi->dgate) {
    edet_launch(k_gate_finish, dim3((in->n * K + 255) / 256), dim3(256), 0, st, a.gate_ws, in->n, K, a.spi, a.sps,
                epi->dgate);
    EDET_LAUNCH_CHECK("edet_pw_bwd(gate finish)");
  }
  return 1;
}
