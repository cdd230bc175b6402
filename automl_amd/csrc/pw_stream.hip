// Streaming pointwise (1x1) convolution kernels for the bf16 path on gfx950 (wave64, MFMA 32x32x16).
//
// The pointwise convolutions of EfficientDet are tall-skinny GEMMs (M = N*H*W rows up to 13.1 M,
// K, N <= a few hundred channels): 25-100 FLOP/B, i.e. HBM-bound on MI355X.  The structure below is
// therefore built around the memory stream, not around the matrix cores:
//
//   * one wave owns a CONTIGUOUS range of 32-row tiles and never synchronises with the other waves of
//     its workgroup inside the main loop (no __syncthreads between loads and MFMAs): 16 waves per CU
//     with independent load -> transform -> MFMA -> store chains hide HBM latency by themselves;
//   * the big operand is read once, in whole rows, 16 bytes per lane ("fixed-column" mapping: lane ->
//     (row r0 + lane / chunks_per_row, channel chunk lane % chunks_per_row), so the per-channel
//     BatchNorm / swish coefficients of the producing layer live in registers), transformed to the
//     activated bf16 value and staged in a wave-private LDS tile; the MFMA B' fragments (8 consecutive
//     k of one row) are ds_read_b128 from there;
//   * the small operand (weights, <= ~100 KB) sits in workgroup-shared LDS for the whole kernel;
//   * the product is computed transposed, D'[out channel][row], so that after the accumulator ->
//     LDS round trip every lane stores 16 contiguous bytes of one row (fully coalesced writes);
//   * BatchNorm statistic partials (sum, sum of squares) are taken from the ROUNDED values that were
//     stored, column-wise from the LDS tile, one partial row per workgroup.
//
// Reference call sites replaced: tf.keras.layers.Conv2D 1x1 in efficientdet/backbone/efficientnet_model.py
// :304-312 (expand), :345-353 (project); efficientdet/tf2/efficientdet_keras.py:286-290 (resample 1x1)
// and the pointwise half of SeparableConv2D (:195-207, :459-464, :546-556).
#include <stdlib.h>

#include "common.h"

namespace pws {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int THREADS = 256;
constexpr int WAVES = 4;
constexpr int TR = 32;        // rows per wave tile
constexpr int NCC = 64;       // output channels per LDS C tile (2 MFMA tiles)
constexpr int PB = 8;         // row passes whose loads are issued back to back

// Fixed-column mapping of a [32 rows][nvec 16-byte chunks] tile onto the 64 lanes of a wave.
struct ColMap {
  int nvec;   // chunks per row
  int rp;     // rows per pass = 64 / nvec   (nvec <= 64)
  int npass;  // ceil(32 / rp)
};

inline ColMap make_colmap(int channels) {
  ColMap m;
  m.nvec = (channels + 7) / 8;
  m.rp = 64 / m.nvec;
  m.npass = (TR + m.rp - 1) / m.rp;
  return m;
}

// LDS row stride (bytes) of a bf16 tile whose rows are read as MFMA fragments (lane = row, 16 B per lane):
// a multiple of 16 B with an ODD number of 16-B slots, so that the 16 rows one ds_read_b128 lane group
// touches fall on 16 different slots of the 64-bank row (a dense 128-B stride is 8-way conflicted:
// SQ_LDS_BANK_CONFLICT was 2x the useful LDS cycles of the 64-channel BiFPN / head layers).
// `need_zero_tail` reserves 16 zero bytes after the row for the k-step that overhangs K (K % 16 == 8).
inline int frag_stride(int cols, bool need_zero_tail) {
  int b = (cols * 2 + 15) / 16 * 16;
  if (need_zero_tail) b += 16;
  if ((b / 16) % 2 == 0) b += 16;
  return b;
}
// stride of the C tile (written 8 B per lane by rows, read 16 B per lane): one 16-B slot of padding
inline int ctile_stride(int cols) { return cols * 2 + 16; }

struct FwdArgs {
  edet_tview_t tv;
  const bf16_t* Wt;   // [N][ldw], k contiguous
  int ldw;
  const float* bias;  // [N] or null
  bf16_t* out;
  int ldo;
  int M, K, N, hw;
  float* stat_partials;
  ColMap ck;
  int G;              // 32-row tiles per super-tile (one super-tile = the loads kept in flight per wave)
  int pst;            // load passes per super-tile (<= NS)
  int SA, SC, SW;     // LDS row strides in bytes
  int Npad;           // N rounded up to 32
  int NWC, nwc;       // weight chunk rows (multiple of 32), number of chunks
  int spw;            // super-tiles per wave (contiguous range)
  int ksteps;         // ceil(K / 16)
  int nvec_out;       // valid 16-byte chunks per output row = ceil(N / 8)
};

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return pack2bf(lo, hi); }

__device__ __forceinline__ void unpack8(const uint4 raw, float x[8]) {
  x[0] = __uint_as_float(raw.x << 16); x[1] = __uint_as_float(raw.x & 0xffff0000u);
  x[2] = __uint_as_float(raw.y << 16); x[3] = __uint_as_float(raw.y & 0xffff0000u);
  x[4] = __uint_as_float(raw.z << 16); x[5] = __uint_as_float(raw.z & 0xffff0000u);
  x[6] = __uint_as_float(raw.w << 16); x[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float x[8]) {
  uint4 o;
  o.x = pack_bf2(x[0], x[1]); o.y = pack_bf2(x[2], x[3]);
  o.z = pack_bf2(x[4], x[5]); o.w = pack_bf2(x[6], x[7]);
  return o;
}

// activated value of 8 raw bf16 elements -> 8 bf16 packed
// OACT: the view's activation is relu / relu6 / hswish (utils.activation_fn, utils.py:36-53; `act` carries the code) --
// a template parameter of the kernels, so that the swish / linear instantiations keep their code and registers
template <bool OACT>
__device__ __forceinline__ uint4 transform8(const uint4 raw, const float sc[8], const float sh[8],
                                            const float gt[8], bool affine, bool swish, bool gate, int act) {
  float x[8];
  unpack8(raw, x);
  if (affine) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], sc[e], sh[e]);
  }
  if (OACT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = act_other_(act, x[e]);
  } else if (swish) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
  }
  if (gate) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] *= gt[e];
  }
  return pack8(x);
}

// D'[out channel][row] for one 32-row tile and one 32-channel slice: A' = weight rows (LDS), B' = staged rows (LDS)
__device__ __forceinline__ f32x16 mma_tile(const unsigned char* wrow, const unsigned char* arow, int ksteps) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int ks = 0; ks < ksteps; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, bf, acc, 0, 0, 0);
  }
  return acc;
}

// the same, continuing an accumulator (a second operand pair contracted into the same D' tile)
__device__ __forceinline__ f32x16 mma_tile_more(const unsigned char* wrow, const unsigned char* arow, int ksteps,
                                                f32x16 acc) {
  for (int ks = 0; ks < ksteps; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, bf, acc, 0, 0, 0);
  }
  return acc;
}

// column sums of a [rows][.] bf16 LDS tile: lane = column `c`
__device__ __forceinline__ void column_stats(const unsigned char* Ct, int SC, int c, int rows_valid, float& s,
                                             float& s2) {
  const bf16_t* colp = reinterpret_cast<const bf16_t*>(Ct) + c;
  const int ld = SC / 2;
  float u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
  if (rows_valid == TR) {
#pragma unroll 4
    for (int r = 0; r < TR; r += 2) {
      const float x0 = bf2f(colp[r * ld]), x1 = bf2f(colp[(r + 1) * ld]);
      u0 += x0; u1 += x1;
      v0 = fmaf(x0, x0, v0); v1 = fmaf(x1, x1, v1);
    }
  } else {
    for (int r = 0; r < rows_valid; ++r) {
      const float x0 = bf2f(colp[r * ld]);
      u0 += x0;
      v0 = fmaf(x0, x0, v0);
    }
  }
  s = u0 + u1;
  s2 = v0 + v1;
}

// ------------------------------------------------------------------------------------ forward
// EXACT: every one of the NS load passes is issued and staged unconditionally (a.pst == NS).  A load under a run-time
// guard makes every wait on the prefetched registers a wait for ALL outstanding loads and stores (r03j, the one-pass
// backward: 30 %); the guarded form remains for the relu / relu6 / hswish instantiations.
// (OACT stays the last template argument: tests/test_gpu_kernels.py reads it off the kernel symbol)
template <int NS, bool EXACT, bool OACT>
__global__ __launch_bounds__(THREADS, NS <= 8 ? 3 : 2) void k_pw_fwd(const FwdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  // LDS carve-up
  unsigned char* Wl = smem;                                        // [NWC][SW]
  float* biasL = reinterpret_cast<float*>(Wl + (size_t)a.NWC * a.SW);   // [Npad]
  float* red = biasL + a.Npad;                                     // [WAVES][2][Npad]: one set of column sums per wave
  const int a_rows = TR * a.G;                                     // rows per super-tile
  const int a_alloc = a.pst * a.ck.rp;                             // >= a_rows: every load pass lands in-bounds
  const size_t wave_bytes = (size_t)a_alloc * a.SA + (size_t)TR * a.SC + (size_t)2 * a.NWC * 4;
  unsigned char* wbase = reinterpret_cast<unsigned char*>(red + 2 * WAVES * a.Npad) + (size_t)wave * wave_bytes;
  unsigned char* At = wbase;                                       // [32*G][SA]
  unsigned char* Ct = wbase + (size_t)a_alloc * a.SA;              // [32][SC]
  float* wst = reinterpret_cast<float*>(Ct + TR * a.SC);           // [2][NWC] this wave's column sums
  const bool want_stats = a.stat_partials != nullptr;

  for (int i = tid; i < a.Npad; i += THREADS) biasL[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
  for (int i = tid; i < 2 * WAVES * a.Npad; i += THREADS) red[i] = 0.f;
  // zero this wave's A buffer once: the bytes after column K stay zero (k-step overhang)
  for (int i = lane; i < a_alloc * a.SA / 16; i += 64) reinterpret_cast<uint4*>(At)[i] = make_uint4(0, 0, 0, 0);

  // K-side mapping and coefficients
  // lanes beyond nvec*rp duplicate the work of lane % (nvec*rp): no divergence, no extra traffic
  const int lane_k = lane % (a.ck.nvec * a.ck.rp);
  const int colK = lane_k % a.ck.nvec, rsub = lane_k / a.ck.nvec;
  const bool activeK = true;
  const bool affine = a.tv.scale != nullptr, swish = a.tv.act == EDET_ACT_SWISH, gated = a.tv.gate != nullptr;
  float sc[8], sh[8], gt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; gt[e] = 1.f; }
  if (affine && activeK) { loadf8(a.tv.scale + colK * 8, sc); loadf8(a.tv.shift + colK * 8, sh); }
  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.tv.data);

  const int nst = (a.M + a_rows - 1) / a_rows;                     // super-tiles
  const int gw = blockIdx.x * WAVES + wave;
  const int st0 = min(nst, gw * a.spw), st1 = min(nst, st0 + a.spw);

  uint4 raw[NS];
  // address = (uniform 64-bit base of the pass) + (32-bit lane offset): one VGPR of addressing for all passes
  const uint32_t lane_off = (uint32_t)(rsub * a.tv.ld + colK * 8) * 2u;
  const size_t pass_bytes = (size_t)a.ck.rp * a.tv.ld * 2;
  auto issue = [&](int st) {
    const int row0 = st * a_rows;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(A) + (size_t)row0 * a.tv.ld * 2;
    const bool full = row0 + a_alloc <= a.M;
    if (EXACT) {
      if (full) {
#pragma unroll
        for (int i = 0; i < NS; ++i) raw[i] = *reinterpret_cast<const uint4*>(base + (size_t)i * pass_bytes + lane_off);
      } else {  // tail: rows past M re-read row M-1 (finite values, never stored)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int r = min(row0 + i * a.ck.rp + rsub, a.M - 1);
          raw[i] = *reinterpret_cast<const uint4*>(A + (size_t)r * a.tv.ld + colK * 8);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i < a.pst) {
          if (full) {
            raw[i] = *reinterpret_cast<const uint4*>(base + (size_t)i * pass_bytes + lane_off);
          } else {
            const int r = min(row0 + i * a.ck.rp + rsub, a.M - 1);
            raw[i] = *reinterpret_cast<const uint4*>(A + (size_t)r * a.tv.ld + colK * 8);
          }
        }
      }
    }
  };

  for (int wc = 0; wc < a.nwc; ++wc) {
    const int n0 = wc * a.NWC;                       // first output channel of this weight chunk
    const int nrows = min(a.NWC, a.Npad - n0);       // multiple of 32
    __syncthreads();
    {  // weight chunk -> LDS (zero-filled beyond N and beyond K)
      const int slots = a.SW / 16;
      for (int q = tid; q < nrows * slots; q += THREADS) {
        const int r = q / slots, s = q - r * slots;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n0 + r < a.N && s * 8 < a.K) v = *reinterpret_cast<const uint4*>(a.Wt + (size_t)(n0 + r) * a.ldw + s * 8);
        *reinterpret_cast<uint4*>(Wl + (size_t)r * a.SW + s * 16) = v;
      }
    }
    __syncthreads();
    for (int i = lane; i < 2 * a.NWC; i += 64) wst[i] = 0.f;

    int gate_img = -1;
    if (st0 < st1) issue(st0);
    for (int st = st0; st < st1; ++st) {
      const int row0 = st * a_rows;
      const int rows_in_st = min(a_rows, a.M - row0);
      // ---- transform the loaded rows into the activated A super-tile
      bool per_row_gate = false;
      if (gated) {
        const int img0 = row0 / a.hw, img1 = (row0 + rows_in_st - 1) / a.hw;
        per_row_gate = img0 != img1;
        if (!per_row_gate && img0 != gate_img && activeK) {
          loadf8(a.tv.gate + (size_t)img0 * a.K + colK * 8, gt);
          gate_img = img0;
        }
        if (per_row_gate) gate_img = -1;
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (EXACT || i < a.pst) {
          const int r = i * a.ck.rp + rsub;
          if (per_row_gate) {
            const uint32_t rr = (uint32_t)min(row0 + r, a.M - 1);
            loadf8(a.tv.gate + (size_t)(rr / (uint32_t)a.hw) * a.K + colK * 8, gt);
          }
          *reinterpret_cast<uint4*>(At + r * a.SA + colK * 16) = transform8<OACT>(raw[i], sc, sh, gt, affine, swish, gated, a.tv.act);
        }
      }
      if (st + 1 < st1) issue(st + 1);               // next super-tile's loads fly during the MFMA phase
      __builtin_amdgcn_wave_barrier();

      for (int sub = 0; sub * TR < rows_in_st; ++sub) {
        const int trow0 = row0 + sub * TR;
        const int rows_valid = min(TR, rows_in_st - sub * TR);
        const unsigned char* arow = At + (size_t)(sub * TR + j) * a.SA + h * 16;
        // ---- MFMA over the output channels of this weight chunk, one C sub-tile (<= 128 channels) at a time
        for (int c0 = 0; c0 < nrows; c0 += NCC) {
          const int ccols = min(NCC, nrows - c0);    // multiple of 32
          for (int nt = 0; nt < ccols / 32; ++nt) {
            const f32x16 acc = mma_tile(Wl + (size_t)(c0 + nt * 32 + j) * a.SW + h * 16, arow, a.ksteps);
            // lane (row j, half h) holds channels nt*32 + (e&3) + 8*(e>>2) + 4*h
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch = c0 + nt * 32 + 8 * g + 4 * h;
              const float4 b4 = *reinterpret_cast<const float4*>(biasL + n0 + ch);
              uint2 pk;
              pk.x = pack_bf2(acc[4 * g + 0] + b4.x, acc[4 * g + 1] + b4.y);
              pk.y = pack_bf2(acc[4 * g + 2] + b4.z, acc[4 * g + 3] + b4.w);
              *reinterpret_cast<uint2*>(Ct + j * a.SC + (nt * 32 + 8 * g + 4 * h) * 2) = pk;
            }
          }
          __builtin_amdgcn_wave_barrier();
          // ---- C sub-tile -> global, 16 bytes per lane in row-major order
          {
            const int nvc_tile = ccols / 8;                                  // chunks per LDS row
            const int nvc = min(nvc_tile, a.nvec_out - (n0 + c0) / 8);       // valid chunks per row
            const int total = rows_valid * nvc;
            int row = lane / nvc, col = lane - row * nvc;
            const int drow = 64 / nvc, dcol = 64 - drow * nvc;
            bf16_t* obase = a.out + (size_t)trow0 * a.ldo + n0 + c0;
#pragma unroll 2
            for (int q = lane; q < total; q += 64) {
              const uint4 v = *reinterpret_cast<const uint4*>(Ct + row * a.SC + col * 16);
              *reinterpret_cast<uint4*>(obase + (size_t)row * a.ldo + col * 8) = v;
              row += drow; col += dcol;
              if (col >= nvc) { col -= nvc; ++row; }
            }
          }
          // ---- statistics of the stored values: lane = column
          if (want_stats) {
            if (lane < ccols) {
              float s, s2;
              column_stats(Ct, a.SC, lane, rows_valid, s, s2);
              wst[c0 + lane] += s;                 // wave-private LDS accumulators
              wst[a.NWC + c0 + lane] += s2;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (want_stats) {
      __builtin_amdgcn_wave_barrier();
      // into this wave's own set (r04: no LDS atomics; the four sets are added in wave order at the end, so the
      // statistics are the same on every run)
      float* redw = red + (size_t)wave * 2 * a.Npad;
      for (int c = lane; c < nrows; c += 64) {
        if (n0 + c < a.N) {
          redw[n0 + c] += wst[c];
          redw[a.Npad + n0 + c] += wst[a.NWC + c];
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    float* dst = a.stat_partials + (size_t)blockIdx.x * 2 * a.N;
    for (int i = tid; i < 2 * a.N; i += THREADS) {
      const int which = i / a.N, col = i - which * a.N;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += red[(size_t)w * 2 * a.Npad + which * a.Npad + col];
      dst[i] = t;
    }
  }
}

// Chooses the weight chunking so that the workgroup's LDS stays under `cap` bytes; returns total bytes
// or 0 when even one 32-row chunk does not fit.
inline size_t plan_lds(int Npad, int a_rows, int SA, int SW, size_t cap, int* NWC_out, int* SC_out) {
  for (int nwcr = Npad; nwcr >= 32; nwcr -= 32) {
    const int ccols = nwcr < NCC ? nwcr : NCC;
    const int SC = ctile_stride(ccols);
    const size_t total = (size_t)nwcr * SW + (size_t)(1 + 2 * WAVES) * Npad * 4 +
                         (size_t)WAVES * ((size_t)a_rows * SA + (size_t)TR * SC + (size_t)2 * nwcr * 4);
    if (total <= cap) {
      *NWC_out = nwcr;
      *SC_out = SC;
      return total;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------ data gradient
// d(in)[m][k] = sum_n dy[m][n] * W[k][n], chained through the input view's activation in the epilogue.
// Same skeleton as the forward kernel with dy as the streamed operand (BatchNorm backward applied on load:
// dy = a*dz + b*y + c, two tensors per row) and an fp32 C tile; the epilogue works on 64-channel chunks with
// a fixed 8-channel column per lane (col = lane & 7, 8 rows per pass), reads the saved conv input x, applies
// act'(z), optionally accumulates into the existing gradient, stores 16 B per lane and keeps the BatchNorm
// backward sums (sum g, sum g*xhat) / the SE dgate sums (sum D*act(z)) in registers across the sub-tiles.
struct BwdArgs {
  edet_gview_t gv;    // dy (contraction length R = gv.c)
  edet_tview_t tv;    // conv input view: raw x, scale, shift, gate, act; KO = tv.c output columns
  const bf16_t* W;    // [KO][ldw], n contiguous
  int ldw;
  edet_bwd_epi_t epi;
  int M, R, KO, hw;
  ColMap cr;
  int G, pst;
  int SA, SC, SW;     // SC: fp32 C tile stride in bytes
  int KOpad;          // KO rounded up to 32
  int spw, ksteps;
  int nvec_out;       // ceil(KO / 8)
};

constexpr int ECC = 64;   // epilogue chunk: channels per C tile

template <int NS, bool GBN, bool OACT>
__global__ __launch_bounds__(THREADS, 2) void k_pw_dgrad(const BwdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  unsigned char* Wl = smem;                                             // [KOpad][SW]
  float* red = reinterpret_cast<float*>(Wl + (size_t)a.KOpad * a.SW);   // [2][KOpad] stats, then [KOpad] gate
  const int a_rows = TR * a.G;
  const int a_alloc = a.pst * a.cr.rp;
  const size_t wave_bytes = (size_t)a_alloc * a.SA + (size_t)TR * a.SC + (size_t)3 * a.KOpad * 4;
  unsigned char* wbase = reinterpret_cast<unsigned char*>(red + 2 * a.KOpad) + (size_t)wave * wave_bytes;
  unsigned char* At = wbase;                                            // [a_alloc][SA] bf16 dy
  unsigned char* Ct = wbase + (size_t)a_alloc * a.SA;                   // [32][SC] fp32
  float* wst = reinterpret_cast<float*>(Ct + TR * a.SC);                // [2][KOpad] stat sums of this wave
  float* wgt = wst + 2 * a.KOpad;                                       // [KOpad] dgate sums of the current image
  const bool want_stats = a.epi.stat_partials != nullptr;
  const bool want_gate = a.epi.dgate != nullptr;      // SE-gated input: the gradient of the gated value is stored as it is
  const bool gate_sums = want_gate && !(a.epi.flags & EDET_EPI_GATE_SUMS_LATER);   // ... and its sums are formed here (atomics)
  const bool swish = !OACT && a.tv.act == EDET_ACT_SWISH, affine = a.tv.scale != nullptr;
  constexpr bool other = OACT;

  for (int i = tid; i < 2 * a.KOpad; i += THREADS) red[i] = 0.f;
  for (int i = lane; i < 3 * a.KOpad; i += 64) wst[i] = 0.f;
  for (int i = lane; i < a_alloc * a.SA / 16; i += 64) reinterpret_cast<uint4*>(At)[i] = make_uint4(0, 0, 0, 0);
  {  // weights -> LDS (zero-filled beyond KO and beyond R)
    const int slots = a.SW / 16;
    for (int q = tid; q < a.KOpad * slots; q += THREADS) {
      const int r = q / slots, sl = q - r * slots;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < a.KO && sl * 8 < a.R) v = *reinterpret_cast<const uint4*>(a.W + (size_t)r * a.ldw + sl * 8);
      *reinterpret_cast<uint4*>(Wl + (size_t)r * a.SW + sl * 16) = v;
    }
  }
  __syncthreads();

  // dy-side mapping
  const int lane_k = lane % (a.cr.nvec * a.cr.rp);
  const int colR = lane_k % a.cr.nvec, rsub = lane_k / a.cr.nvec;
  const int kvalid = min(8, a.R - colR * 8);                           // elements of this chunk inside R
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gv.dz);
  const bf16_t* Y = reinterpret_cast<const bf16_t*>(a.gv.y);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.tv.data);
  bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);
  const uint32_t lane_off = (uint32_t)(rsub * a.gv.ld + colR * 8) * 2u;
  const size_t pass_bytes = (size_t)a.cr.rp * a.gv.ld * 2;

  const int nst = (a.M + a_rows - 1) / a_rows;
  const int gw = blockIdx.x * WAVES + wave;
  const int st0 = min(nst, gw * a.spw), st1 = min(nst, st0 + a.spw);

  uint4 rz[NS], ry[GBN ? NS : 1];
  auto issue = [&](int st) {
    const int row0 = st * a_rows;
    const bool full = row0 + a_alloc <= a.M;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i < a.pst) {
        size_t off;
        if (full) off = (size_t)row0 * a.gv.ld * 2 + (size_t)i * pass_bytes + lane_off;
        else off = ((size_t)min(row0 + i * a.cr.rp + rsub, a.M - 1) * a.gv.ld + colR * 8) * 2;
        rz[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(DZ) + off);
        if (GBN) ry[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(Y) + off);
      }
    }
  };

  // epilogue mapping: 64-channel chunk, 8 lanes per row
  const int ecol = lane & 7, erow = lane >> 3;
  int gate_img = -1;                       // image whose dgate sums are in wgt
  if (st0 < st1) issue(st0);
  for (int st = st0; st < st1; ++st) {
    const int row0 = st * a_rows;
    const int rows_in_st = min(a_rows, a.M - row0);
    {  // dy = a*dz + b*y + c, masked to the R valid columns, as bf16 into the A super-tile
      float ga[8], gb[8], gc[8];
      if (GBN) {
        loadf8(a.gv.a + colR * 8, ga); loadf8(a.gv.b + colR * 8, gb); loadf8(a.gv.cc + colR * 8, gc);
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i < a.pst) {
          float x[8];
          unpack8(rz[i], x);
          if (GBN) {
            float y[8];
            unpack8(ry[i], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(ga[e], x[e], fmaf(gb[e], y[e], gc[e]));
          }
          if (kvalid < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e >= kvalid) x[e] = 0.f;
          }
          *reinterpret_cast<uint4*>(At + (i * a.cr.rp + rsub) * a.SA + colR * 16) = pack8(x);
        }
      }
    }
    if (st + 1 < st1) issue(st + 1);
    __builtin_amdgcn_wave_barrier();

    // dgate bookkeeping: sums in wgt belong to one image; a super-tile that straddles two images goes
    // straight to global atomics
    bool gate_direct = false;
    if (gate_sums) {
      const int img0 = row0 / a.hw, img1 = (row0 + rows_in_st - 1) / a.hw;
      gate_direct = img0 != img1;
      if (gate_img >= 0 && (gate_direct || img0 != gate_img)) {
        for (int c = lane; c < a.KO; c += 64) {
          const float v = wgt[c];
          if (v != 0.f) atomicAdd(&a.epi.dgate[(size_t)gate_img * a.KO + c], v);
          wgt[c] = 0.f;
        }
        gate_img = -1;
      }
      if (!gate_direct) gate_img = img0;
      __builtin_amdgcn_wave_barrier();
    }

    for (int c0 = 0; c0 < a.KOpad; c0 += ECC) {
      const int ccols = min(ECC, a.KOpad - c0);                // 32 or 64
      const int ch0 = c0 + ecol * 8;                           // this lane's 8 output channels
      const bool col_ok = ch0 < a.KO && ecol * 8 < ccols;
      // s1: sum g (stats) or sum D*act(z) (gate); s2: sum g*x, turned into sum g*xhat when it is flushed
      float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; s1[e] = s2[e] = 0.f; }
      if (col_ok && affine) { loadf8(a.tv.scale + ch0, sc); loadf8(a.tv.shift + ch0, sh); }
      const bool need_x = swish || other || want_gate || want_stats;
      for (int sub = 0; sub * TR < rows_in_st; ++sub) {
        const int trow0 = row0 + sub * TR;
        const int rows_valid = min(TR, rows_in_st - sub * TR);
        // saved conv input for this (sub-tile, chunk): 4 passes of 8 rows, in flight during the MFMAs
        uint4 xr[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          xr[p] = make_uint4(0, 0, 0, 0);
          const int r = p * 8 + erow;
          if (need_x && col_ok && r < rows_valid)
            xr[p] = *reinterpret_cast<const uint4*>(X + (size_t)(trow0 + r) * a.tv.ld + ch0);
        }
        const unsigned char* arow = At + (size_t)(sub * TR + j) * a.SA + h * 16;
        for (int nt = 0; nt < ccols / 32; ++nt) {
          const f32x16 acc = mma_tile(Wl + (size_t)(c0 + nt * 32 + j) * a.SW + h * 16, arow, a.ksteps);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(Ct + j * a.SC + (nt * 32 + 8 * g + 4 * h) * 4) =
                make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = p * 8 + erow;
          if (col_ok && r < rows_valid) {
            const float4 d0 = *reinterpret_cast<const float4*>(Ct + r * a.SC + ecol * 32);
            const float4 d1 = *reinterpret_cast<const float4*>(Ct + r * a.SC + ecol * 32 + 16);
            float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float x[8], g[8];
            unpack8(xr[p], x);
            const size_t off = (size_t)(trow0 + r) * a.tv.ld + ch0;
            if (want_gate && !gate_sums) {
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] = d[e];
            } else if (want_gate) {
              float gsum[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float z = fmaf(x[e], sc[e], sh[e]);
                gsum[e] = d[e] * (other ? act_other_(a.tv.act, z) : (swish ? swishf_(z) : z));
                g[e] = d[e];
              }
              if (gate_direct) {
                const int img = (trow0 + r) / a.hw;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (ch0 + e < a.KO) atomicAdd(&a.epi.dgate[(size_t)img * a.KO + ch0 + e], gsum[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[e] += gsum[e];
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
            if (a.epi.beta) {
              float old[8];
              unpack8(*reinterpret_cast<const uint4*>(GO + off), old);
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] += old[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) if (ch0 + e >= a.KO) g[e] = 0.f;
            *reinterpret_cast<uint4*>(GO + off) = pack8(g);
            if (want_stats) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                s1[e] += g[e];
                s2[e] = fmaf(g[e], x[e], s2[e]);
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      // chunk sums -> wave LDS accumulators (8 lanes share a column: LDS atomics)
      if (col_ok) {
        if (want_stats) {
          float mu[8], rs[8];
          loadf8(a.epi.mean + ch0, mu);
          loadf8(a.epi.rstd + ch0, rs);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (ch0 + e < a.KO) {
              atomicAdd(&wst[ch0 + e], s1[e]);
              atomicAdd(&wst[a.KOpad + ch0 + e], rs[e] * (s2[e] - mu[e] * s1[e]));   // sum g*(x-mean)*rstd
            }
          }
        }
        if (gate_sums && !gate_direct) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (ch0 + e < a.KO) atomicAdd(&wgt[ch0 + e], s1[e]);
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (gate_sums && gate_img >= 0) {
    for (int c = lane; c < a.KO; c += 64) {
      const float v = wgt[c];
      if (v != 0.f) atomicAdd(&a.epi.dgate[(size_t)gate_img * a.KO + c], v);
    }
  }
  if (want_stats) {
    for (int c = lane; c < a.KO; c += 64) {
      atomicAdd(&red[c], wst[c]);
      atomicAdd(&red[a.KOpad + c], wst[a.KOpad + c]);
    }
    __syncthreads();
    float* dst = a.epi.stat_partials + (size_t)blockIdx.x * 2 * a.KO;
    for (int i = tid; i < 2 * a.KO; i += THREADS) {
      const int which = i / a.KO, col = i - which * a.KO;
      dst[i] = red[which * a.KOpad + col];
    }
  }
}

// ---------------------------------------------------------------------------- weight gradient
// dW[k][n] = sum_m view(in)[m][k] * dy[m][n]: the contraction runs over the rows, so both operands are
// needed "transposed" (8 consecutive rows of one channel per lane).  Each wave owns a 64 x CV block of dW
// (U = the operand with more channels, sliced 64 wide; V = the other one, CV <= 64) over a contiguous row
// range: per 32-row step it loads its U slice and V in whole 16-byte chunks (fixed column per lane, so the
// BatchNorm / swish / BatchNorm-backward coefficients stay in registers), stages the transformed bf16 rows in
// wave-private LDS and gathers the MFMA fragments column-wise with ds_read_u16 (64 B/clk/CU, six times the
// HBM feed rate).  Partial blocks go to a workspace [split][K][N] that a second kernel sums into dW:
// deterministic, no atomics.
struct WgArgs {
  edet_tview_t tv;    // conv input view, K = tv.c
  edet_gview_t gv;    // dy, N = gv.c
  float* ws;          // [S][K][N]
  int M, K, N, hw;
  int CU, CV;         // channels of U and V
  int nus;            // 64-wide slices of U
  int S;              // row splits
  int rows_per_split; // multiple of 32
  int cpwV;           // V-side lanes per row (power of two >= CV/8, <= 8)
};

template <bool IS_G, bool GBN>
struct OperandRegs {
  uint4 r[4];
  uint4 y[(IS_G && GBN) ? 4 : 1];
};

// loads pass p (rows rbase + p*rpp + rowl) of one operand
template <bool IS_G, bool GBN>
__device__ __forceinline__ void wg_issue(const WgArgs& a, OperandRegs<IS_G, GBN>& o, int npass, int rpp, int rowl,
                                         int ch, bool col_ok, int m0, int m_end) {
  const bf16_t* P = reinterpret_cast<const bf16_t*>(IS_G ? a.gv.dz : a.tv.data);
  const bf16_t* Y = reinterpret_cast<const bf16_t*>(a.gv.y);
  const int ld = IS_G ? a.gv.ld : a.tv.ld;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p < npass) {
      const int m = m0 + p * rpp + rowl;
      o.r[p] = make_uint4(0, 0, 0, 0);
      if (IS_G && GBN) o.y[p] = make_uint4(0, 0, 0, 0);
      if (col_ok && m < m_end) {
        o.r[p] = *reinterpret_cast<const uint4*>(P + (size_t)m * ld + ch);
        if (IS_G && GBN) o.y[p] = *reinterpret_cast<const uint4*>(Y + (size_t)m * ld + ch);
      }
    }
  }
}

// transforms the loaded passes and writes them (bf16) into the operand's LDS tile [32][stride]
template <bool IS_G, bool GBN, bool OACT>
__device__ __forceinline__ void wg_stage(const WgArgs& a, const OperandRegs<IS_G, GBN>& o, int npass, int rpp,
                                         int rowl, int coll, int ch, int cmax, bool col_ok, int m0, int m_end,
                                         unsigned char* tile, int stride) {
  if (!col_ok) return;
  float c0[8], c1[8], c2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { c0[e] = 1.f; c1[e] = 0.f; c2[e] = 0.f; }
  if (IS_G) {
    if (GBN) { loadf8(a.gv.a + ch, c0); loadf8(a.gv.b + ch, c1); loadf8(a.gv.cc + ch, c2); }
  } else if (a.tv.scale) {
    loadf8(a.tv.scale + ch, c0);
    loadf8(a.tv.shift + ch, c1);
  }
  const int valid = min(8, cmax - ch);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p < npass) {
      const int r = p * rpp + rowl;
      const int m = m0 + r;
      float x[8];
      unpack8(o.r[p], x);
      if (m < m_end) {
        if (IS_G) {
          if (GBN) {
            float y[8];
            unpack8(o.y[p], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(c0[e], x[e], fmaf(c1[e], y[e], c2[e]));
          }
        } else {
          if (a.tv.scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], c0[e], c1[e]);
          }
          if (OACT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = act_other_(a.tv.act, x[e]);
          } else if (a.tv.act == EDET_ACT_SWISH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
          }
          if (a.tv.gate) {
            float gt[8];
            loadf8(a.tv.gate + (size_t)(m / a.hw) * a.K + ch, gt);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= gt[e];
          }
        }
        if (valid < 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e >= valid) x[e] = 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      if (r < TR) *reinterpret_cast<uint4*>(tile + r * stride + coll * 16) = pack8(x);
    }
  }
}

// 8 consecutive rows (8*h + 16*ks ...) of column `col` of a [32][stride] bf16 tile
__device__ __forceinline__ bf16x8 column_frag(const unsigned char* tile, int stride, int row0, int col) {
  const bf16_t* p = reinterpret_cast<const bf16_t*>(tile + row0 * stride) + col;
  const int ld = stride / 2;
  s16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (short)p[e * ld];
  return __builtin_bit_cast(bf16x8, v);
}

// The same fragment through the gfx950 transpose read (ds_read_b64_tr_b16): within a 16-lane group lane i supplies the
// address of 4 consecutive channels (8 bytes) of row row0 + i / 4 -- together a [4 rows][16 channels] block starting at
// channel col0 -- and receives the 4 rows of channel col0 + i.  Two reads (rows +0..3, +4..7) replace eight 2-byte
// reads and their packing.  Addresses are 8-byte aligned: 16-byte row strides, col0 % 16 == 0.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v_t;
__device__ __forceinline__ bf16x8 column_frag_tr(const unsigned char* tile, int stride, int row0, int col0, int fi) {
  const unsigned char* p = tile + (size_t)(row0 + (fi >> 2)) * stride + (col0 + (fi & 3) * 4) * 2;
  typedef __attribute__((address_space(3))) bf16x4v_t* lds_ptr_t;
  const bf16x4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p));
  const bf16x4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p + 4 * stride));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int WG_SU = 64 * 2 + 16;   // U tile stride (64 channels)
constexpr int WG_SV = 64 * 2 + 16;   // V tile stride (up to 64 channels)

template <bool UG, bool GBN, bool OACT>   // UG: U is the gradient operand (N >= K)
__global__ __launch_bounds__(THREADS, 2) void k_pw_wgrad(const WgArgs a) {
  __shared__ __align__(16) unsigned char smem[WAVES * TR * (WG_SU + WG_SV)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned char* Ut = smem + wave * TR * (WG_SU + WG_SV);
  unsigned char* Vt = Ut + TR * WG_SU;
  // wave -> (U slice, row split)
  int us, rs;
  if (a.nus >= 3) {
    const int nus4 = (a.nus + 3) / 4;
    us = (blockIdx.x % nus4) * 4 + wave;
    rs = blockIdx.x / nus4;
  } else {
    us = wave % a.nus;
    rs = blockIdx.x * (WAVES / a.nus) + wave / a.nus;
  }
  if (us >= a.nus || rs >= a.S) return;
  const int m_begin = rs * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);

  // U side: 8 chunks per row, 8 rows per pass, 4 passes; V side: cpwV chunks per row
  const int ucol = lane & 7, urow = lane >> 3;
  const int uch = us * 64 + ucol * 8;
  const bool u_ok = uch < a.CU;
  const int vcol = lane & (a.cpwV - 1), vrow = lane / a.cpwV;
  const int rppV = 64 / a.cpwV;
  const int npassV = rppV >= TR ? 1 : TR / rppV;
  const int vch = vcol * 8;
  const bool v_ok = vch < a.CV && vrow < TR;
  const int nvt = (a.CV + 31) / 32;                 // 1 or 2 V tiles

  // zero both tiles once (columns never written stay zero)
  for (int i = lane; i < TR * (WG_SU + WG_SV) / 16; i += 64) reinterpret_cast<uint4*>(Ut)[i] = make_uint4(0, 0, 0, 0);

  f32x16 acc[2][2];
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int vt = 0; vt < 2; ++vt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ut][vt][e] = 0.f;

  OperandRegs<UG, GBN> ou;
  OperandRegs<!UG, GBN> ov;
  if (m_begin < m_end) {
    wg_issue<UG, GBN>(a, ou, 4, 8, urow, uch, u_ok, m_begin, m_end);
    wg_issue<!UG, GBN>(a, ov, npassV, rppV, vrow, vch, v_ok, m_begin, m_end);
  }
  const int j = lane & 31, h = lane >> 5;
  for (int m0 = m_begin; m0 < m_end; m0 += TR) {
    wg_stage<UG, GBN, OACT>(a, ou, 4, 8, urow, ucol, uch, a.CU, u_ok, m0, m_end, Ut, WG_SU);
    wg_stage<!UG, GBN, OACT>(a, ov, npassV, rppV, vrow, vcol, vch, a.CV, v_ok, m0, m_end, Vt, WG_SV);
    if (m0 + TR < m_end) {
      wg_issue<UG, GBN>(a, ou, 4, 8, urow, uch, u_ok, m0 + TR, m_end);
      wg_issue<!UG, GBN>(a, ov, npassV, rppV, vrow, vch, v_ok, m0 + TR, m_end);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 16 + h * 8;
      const bf16x8 u0 = column_frag(Ut, WG_SU, r0, j);
      const bf16x8 u1 = column_frag(Ut, WG_SU, r0, 32 + j);
      const bf16x8 v0 = column_frag(Vt, WG_SV, r0, j);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0, v0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1, v0, acc[1][0], 0, 0, 0);
      if (nvt > 1) {
        const bf16x8 v1 = column_frag(Vt, WG_SV, r0, 32 + j);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0, v1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1, v1, acc[1][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // D[i = U channel][j = V channel]: lane holds V channel j (+32*vt), U channels (e&3)+8*(e>>2)+4*h (+32*ut)
  float* dst = a.ws + (size_t)rs * a.K * a.N;
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int vt = 0; vt < 2; ++vt) {
      if (vt < nvt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int iu = us * 64 + ut * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int jv = vt * 32 + j;
          if (iu < a.CU && jv < a.CV) {
            const int k = UG ? jv : iu, n = UG ? iu : jv;
            dst[(size_t)k * a.N + n] = acc[ut][vt][e];
          }
        }
      }
    }
}

// ------------------------------------------------------------------ fused data + weight gradient
// Both gradients of a pointwise convolution in ONE pass over (dz, y, x): the data-gradient kernel above and the
// weight-gradient kernel each stream the gradient pair (2 N channels per row) and the saved input (K channels),
// i.e. 7 tensor streams where one pass needs 4 (r01h PMC: 329 + 363 MB per launch pair against 194 MB
// algorithmic each).  Here a wave owns a contiguous range of 32-row tiles; per tile
//   * dz, y AND x are prefetched one tile ahead in whole 16-byte chunks (fixed column per lane: the BatchNorm
//     backward coefficients stay in registers), dy = a*dz + b*y + c goes to the LDS tile Dt as bf16, x is parked raw
//     in the LDS tile Xt;
//   * D'[k][row] = W[k][:] . dy[row][:] on 32x32x16 MFMAs, through the fp32 C tile into the 8-channel-per-lane
//     epilogue of k_pw_dgrad (act'(z), accumulate, BatchNorm-backward sums, SE dgate sums), which takes x from Xt
//     and leaves the ACTIVATED operand act(bn(x)) * gate there as bf16;
//   * dW[k][n] += sum_rows xa[row][k] * dy[row][n] on 16x16x32 MFMAs whose contraction runs over the tile's 32
//     rows: fragments are gathered column-wise (8 consecutive rows of one channel per lane) from Xt and Dt; the dW
//     block (<= 4 x 9 tiles of 16 x 16) stays in registers for the whole kernel.
// One wave per SIMD (the accumulators and a whole tile of loads in flight take ~400 registers): latency is hidden
// by the 14-21 KB every wave keeps in flight, not by occupancy.  At the end the four waves of a workgroup add their
// dW blocks in LDS (fixed order) and write one partial per workgroup; edet_reduce_partials sums them.
//
// NOY (r03): the BatchNorm backward WITHOUT the saved convolution output.  dy = a*dz + b*y + c and y = x~ W (x~ the
// operand the forward fed the matrix cores), so
//     dx~ = dy W^T   = dz (W diag(a))^T + x~ (W diag(b) W^T) + c W^T          = dz Wa^T + x~ G + v
//     dW  = x~^T dy  = (x~^T dz) diag(a) + (x~^T x~) W diag(b) + (sum_p x~) c^T = P diag(a) + S W diag(b) + s c^T
// with G = W diag(b) W^T a KO x KO matrix and S = x~^T x~ the Gram matrix of the input (KO <= 32 here): the 6x wider
// y is never read (the kernel's traffic drops from 2N + 2K to N + 2K channels per row: -43 % for 16 -> 96), dz goes
// from HBM to the LDS operand tile without being unpacked, and the only extra matrix work is one k-step against G
// and one or four 16 x 16 tiles of S.  Wa, G and v are made in the prologue from the weights already in LDS; the
// partial written per workgroup is [P | S | s] and k_noy_apply finishes dW.  Used when the convolution input is a
// plain stored tensor (every MBConv expansion reads a block output), so x~ = x and there is no chain epilogue work.
struct FusedArgs {
  edet_gview_t gv;    // dy: R = gv.c channels (the convolution's output channels)
  edet_tview_t tv;    // conv input view: raw x, scale, shift, gate, act; KO = tv.c channels
  const bf16_t* W;    // [KO][ldw], n contiguous (compute copy of the kernel)
  int ldw;
  edet_bwd_epi_t epi;
  float* ws;          // [grid][KO][R] fp32 partial weight gradients
  int M, R, KO, hw;
  ColMap cr, cx;      // load mappings of dy and of x
  int SA, SX, SC, SW; // LDS row strides in bytes: Dt, Xt, C tile, weights
  int KOpad;          // KO rounded up to 32
  int TK, TN;         // 16-wide tiles of dW along k and n
  int tpw, ksteps;    // steps per wave; ceil(R / 16)
  int SG, kxsteps;    // NOY: LDS row stride of G in bytes, ceil(KOpad / 16)
  int G;              // 32-row tiles per step: the loads of a whole step (32 G rows of dz, y, x) are in flight at once
};

// FT_S x FT_L = 16 x 16 tiles of dW kept per wave (min(TK, TN) <= FT_S, max(TK, TN) <= FT_L): 2 x 9 covers the expand
// layers (16..32 input channels, up to 144 output channels), 3 x 9 the project layers (up to 144 -> 48), 4 x 4 the
// 64 -> 64 layers of the BiFPN and the heads.
// NSR / NSX: load passes per step of dy / x, all issued unconditionally (r03j: run-time guards around the loads cost
// 30 % -- the waits degrade to vmcnt(0)); the host picks the instantiation that fits the step with the fewest rows
// to spare.  The round-2 kernel had two pairs, (8, 1) and (12, 2): 8 passes of 32 rows each for a 16-channel gradient
// whose 32-row tile needs one -- the reason the project layers measured slower in it.
// NCH: 64-channel epilogue chunks (KOpad <= 64 NCH); the per-channel sums (BatchNorm backward, SE gate gradient)
// are carried in registers across all tiles of the wave and reach LDS once per kernel (or once per image).
// BETA: the instantiation can accumulate into gout (epi.beta; the old gradient rides along with x)
// STRADDLE: a 32-row tile may lie in two images (hw % 32 != 0) while an SE gate / gate gradient is involved
template <int NSR, int NSX, bool GBN, int FT_S, int FT_L, bool NOY = false, int NCH = 1, bool BETA = true, bool STRADDLE = true>
__global__ __launch_bounds__(THREADS, 1) void k_pw_bwd_fused(const FusedArgs a) {
  static_assert(!NOY || GBN, "NOY is a form of the BatchNorm backward on load");
  static_assert(!NOY || NCH == 1, "NOY: at most 32 input channels");
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const bool has_beta = BETA && a.epi.beta != 0;
  const int srows = TR * a.G;                                           // rows per step
  unsigned char* Wl = smem;                                             // [KOpad][SW]
  float* red = reinterpret_cast<float*>(Wl + (size_t)a.KOpad * a.SW);   // [2][KOpad] stats
  float* coefS = red + 2 * a.KOpad;                                     // [KOpad] scale, [KOpad] shift of the view
  float* coefH = coefS + a.KOpad;
  const size_t wave_bytes = (size_t)srows * a.SA + (size_t)(has_beta ? 2 : 1) * srows * a.SX +
                            (size_t)TR * a.SC + (size_t)4 * a.KOpad * 4;
  float* vL = coefH + a.KOpad;                                          // NOY: [KOpad] v = c W^T
  unsigned char* Gl = reinterpret_cast<unsigned char*>(vL + (NOY ? a.KOpad : 0));      // NOY: [KOpad][SG] bf16 G
  unsigned char* wbase = Gl + (NOY ? (size_t)a.KOpad * a.SG : 0) + (size_t)wave * wave_bytes;
  unsigned char* Dt = wbase;                                            // [srows][SA] bf16 dy
  unsigned char* Xt = Dt + (size_t)srows * a.SA;                        // [srows][SX] bf16 x, then act(x)
  unsigned char* Ot = Xt + (size_t)srows * a.SX;                        // [srows][SX] bf16 old gout (beta only)
  unsigned char* Ct = Ot + (has_beta ? (size_t)srows * a.SX : 0);       // [32][SC] fp32
  float* wst = reinterpret_cast<float*>(Ct + TR * a.SC);                // [2][KOpad] sums (g, g*x) of this wave
  float* wgt = wst + 2 * a.KOpad;                                       // [KOpad] dgate sums of one image
  float* gateL = wgt + a.KOpad;                                         // [KOpad] SE gate of the current image
  // NOY: the host guarantees a plain input view and no sums (compile-time false: the code is not generated)
  const bool want_stats = !NOY && a.epi.stat_partials != nullptr;
  const bool want_gate = !NOY && a.epi.dgate != nullptr;
  const bool swish = !NOY && a.tv.act == EDET_ACT_SWISH, affine = !NOY && a.tv.scale != nullptr;
  const bool gated = !NOY && a.tv.gate != nullptr;

  // Nothing inside the tile loop may wait on a global load other than the prefetched step (vmcnt is in order: a
  // wait for a later small load would drain the whole prefetch): per-channel vectors live in LDS.
  for (int i = tid; i < 2 * a.KOpad; i += THREADS) red[i] = 0.f;
  for (int i = tid; i < a.KOpad; i += THREADS) {
    coefS[i] = (affine && i < a.KO) ? a.tv.scale[i] : 1.f;
    coefH[i] = (affine && i < a.KO) ? a.tv.shift[i] : 0.f;
  }
  for (int i = lane; i < 3 * a.KOpad; i += 64) wst[i] = 0.f;
  for (int i = lane; i < a.KOpad; i += 64) gateL[i] = 1.f;
  for (int i = lane; i < (int)(((size_t)srows * a.SA + (size_t)srows * a.SX) / 16); i += 64)
    reinterpret_cast<uint4*>(Dt)[i] = make_uint4(0, 0, 0, 0);
  {  // weights -> LDS (zero-filled beyond KO and beyond R)
    const int slots = a.SW / 16;
    for (int q = tid; q < a.KOpad * slots; q += THREADS) {
      const int r = q / slots, sl = q - r * slots;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < a.KO && sl * 8 < a.R) v = *reinterpret_cast<const uint4*>(a.W + (size_t)r * a.ldw + sl * 8);
      *reinterpret_cast<uint4*>(Wl + (size_t)r * a.SW + sl * 16) = v;
    }
  }
  __syncthreads();
  if (NOY) {
    // G[k][k'] = sum_n W[k][n] b[n] W[k'][n] (bf16 MFMA operand, zero beyond KO), v[k] = sum_n c[n] W[k][n] -- from the
    // bf16 weights the forward used; then the LDS weights become Wa[k][n] = a[n] W[k][n]
    for (int q = tid; q < a.KOpad * a.KOpad; q += THREADS) {
      const int k = q / a.KOpad, k2 = q - k * a.KOpad;
      float g = 0.f;
      if (k < a.KO && k2 < a.KO) {
        const bf16_t* w1 = reinterpret_cast<const bf16_t*>(Wl + (size_t)k * a.SW);
        const bf16_t* w2 = reinterpret_cast<const bf16_t*>(Wl + (size_t)k2 * a.SW);
        for (int n = 0; n < a.R; ++n) g = fmaf(bf2f(w1[n]) * a.gv.b[n], bf2f(w2[n]), g);
      }
      reinterpret_cast<bf16_t*>(Gl + (size_t)k * a.SG)[k2] = f2bf(g);
    }
    for (int q = tid; q < a.KOpad * (a.SG / 2 - a.KOpad); q += THREADS) {      // zero tail of every G row
      const int k = q / (a.SG / 2 - a.KOpad), t = q - k * (a.SG / 2 - a.KOpad);
      reinterpret_cast<bf16_t*>(Gl + (size_t)k * a.SG)[a.KOpad + t] = 0;
    }
    for (int k = tid; k < a.KOpad; k += THREADS) {
      float t = 0.f;
      if (k < a.KO) {
        const bf16_t* w1 = reinterpret_cast<const bf16_t*>(Wl + (size_t)k * a.SW);
        for (int n = 0; n < a.R; ++n) t = fmaf(a.gv.cc[n], bf2f(w1[n]), t);
      }
      vL[k] = t;
    }
    __syncthreads();
    for (int q = tid; q < a.KO * a.R; q += THREADS) {
      const int k = q / a.R, n = q - k * a.R;
      bf16_t* wp = reinterpret_cast<bf16_t*>(Wl + (size_t)k * a.SW) + n;
      *wp = f2bf(bf2f(*wp) * a.gv.a[n]);
    }
    __syncthreads();
  }

  // load mappings (lanes beyond nvec*rp duplicate the work of lane % (nvec*rp): no divergence)
  const int lane_r = lane % (a.cr.nvec * a.cr.rp);
  const int colR = lane_r % a.cr.nvec, rsubR = lane_r / a.cr.nvec;
  const int kvalid = min(8, a.R - colR * 8);
  const int lane_x = lane % (a.cx.nvec * a.cx.rp);
  const int colX = lane_x % a.cx.nvec, rsubX = lane_x / a.cx.nvec;
  const unsigned char* DZ = reinterpret_cast<const unsigned char*>(a.gv.dz);
  const unsigned char* Y = reinterpret_cast<const unsigned char*>(a.gv.y);
  const unsigned char* X = reinterpret_cast<const unsigned char*>(a.tv.data);
  bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);

  const int nstep = (a.M + srows - 1) / srows;
  const int gw = blockIdx.x * WAVES + wave;
  const int t0 = min(nstep, gw * a.tpw), t1 = min(nstep, t0 + a.tpw);

  // native vector types: a HIP uint4 struct copied whole from a register array to LDS is not promoted to registers
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rz[NSR], ry[(GBN && !NOY) ? NSR : 1];
  u32x4 rx[NSX], ro[BETA ? NSX : 1];
  // per-lane byte offsets inside a step (loop invariant) and the uniform byte strides between load passes: a step's
  // addresses are one scalar base + these (r03l: the 64-bit multiply-add per load and lane was ~15 % of the VALU work
  // of the expand layers)
  const uint32_t voffR = (uint32_t)(rsubR * a.gv.ld + colR * 8) * 2, voffX = (uint32_t)(rsubX * a.tv.ld + colX * 8) * 2;
  const uint32_t passR = (uint32_t)a.cr.rp * a.gv.ld * 2, passX = (uint32_t)a.cx.rp * a.tv.ld * 2;
  const int rows_touched = max(NSR * a.cr.rp, NSX * a.cx.rp);
  auto issue = [&](int t) {
    // every instantiated pass is issued, unconditionally and back to back (a load under a branch makes the compiler
    // wait for ALL outstanding loads and stores wherever one of them is consumed); a pass that reaches beyond the step
    // reads rows of the wave's NEXT step (they are in L2 when that step asks for them), rows past M re-read row M-1
    const int row0 = t * srows;
    if (row0 + rows_touched <= a.M) {
      const unsigned char* bz = DZ + (size_t)row0 * a.gv.ld * 2;
      const unsigned char* by = Y + (size_t)row0 * a.gv.ld * 2;
      const unsigned char* bx = X + (size_t)row0 * a.tv.ld * 2;
      const unsigned char* bo = reinterpret_cast<const unsigned char*>(GO) + (size_t)row0 * a.tv.ld * 2;
#pragma unroll
      for (int i = 0; i < NSR; ++i) {
        rz[i] = *reinterpret_cast<const u32x4*>(bz + (size_t)i * passR + voffR);
        if (GBN && !NOY) ry[i] = *reinterpret_cast<const u32x4*>(by + (size_t)i * passR + voffR);
      }
#pragma unroll
      for (int i = 0; i < NSX; ++i) rx[i] = *reinterpret_cast<const u32x4*>(bx + (size_t)i * passX + voffX);
      if (has_beta) {       // the gradient already in gout (residual / second consumer) rides along with x
#pragma unroll
        for (int i = 0; i < NSX; ++i) ro[BETA ? i : 0] = *reinterpret_cast<const u32x4*>(bo + (size_t)i * passX + voffX);
      }
    } else {                // the last steps of the tensor: clamp every row
#pragma unroll
      for (int i = 0; i < NSR; ++i) {
        const size_t off = ((size_t)min(row0 + i * a.cr.rp + rsubR, a.M - 1) * a.gv.ld + colR * 8) * 2;
        rz[i] = *reinterpret_cast<const u32x4*>(DZ + off);
        if (GBN && !NOY) ry[i] = *reinterpret_cast<const u32x4*>(Y + off);
      }
#pragma unroll
      for (int i = 0; i < NSX; ++i) {
        const size_t off = ((size_t)min(row0 + i * a.cx.rp + rsubX, a.M - 1) * a.tv.ld + colX * 8) * 2;
        rx[i] = *reinterpret_cast<const u32x4*>(X + off);
      }
      if (has_beta) {
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
          const size_t off = ((size_t)min(row0 + i * a.cx.rp + rsubX, a.M - 1) * a.tv.ld + colX * 8) * 2;
          ro[BETA ? i : 0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(GO) + off);
        }
      }
    }
  };

  // dW block of this wave: acc[s][l] is the 16 x 16 tile (s, l) of the (small side, large side) tile grid
  const bool ksmall = a.TK <= a.TN;
  const int TS = ksmall ? a.TK : a.TN, TL = ksmall ? a.TN : a.TK;
  f32x4 acc[FT_S][FT_L];
#pragma unroll
  for (int s = 0; s < FT_S; ++s)
#pragma unroll
    for (int l = 0; l < FT_L; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][l][e] = 0.f;

  float ga[8], gb[8], gc[8];
  if (GBN && !NOY) {
    loadf8(a.gv.a + colR * 8, ga); loadf8(a.gv.b + colR * 8, gb); loadf8(a.gv.cc + colR * 8, gc);
  }
  // this lane's running column sums over all its tiles (its 8 channels of every epilogue chunk never change):
  // ra = sum g (BatchNorm backward) or sum d*act (SE gate gradient, per image), rb = sum g*x; NOY: xacc = sum x
  float ra[NCH][8], rb[NCH][8];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci)
#pragma unroll
    for (int e = 0; e < 8; ++e) ra[ci][e] = rb[ci][e] = 0.f;
  float xacc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) xacc[e] = 0.f;
  // NOY: the Gram matrix S = x^T x of this wave's rows, 16 x 16 tiles (s, s2) of the k side
  f32x4 accS[NOY ? FT_S : 1][NOY ? FT_S : 1];
#pragma unroll
  for (int s = 0; s < (NOY ? FT_S : 1); ++s)
#pragma unroll
    for (int s2 = 0; s2 < (NOY ? FT_S : 1); ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) accS[s][s2][e] = 0.f;
  // epilogue mapping of the 64-channel chunk ci: lanes per row = the chunk's 16-byte column groups rounded up to a
  // power of two (a 16-channel input: 2 lanes per row, all 32 rows of the tile in ONE pass instead of four passes with
  // 6 of 8 lanes idle -- the wave issues every instruction of a pass whatever the number of live lanes)
  auto chunk_lsh = [&](int ci) {
    const int nv = min(8, (a.KO - ci * ECC + 7) / 8);
    return nv <= 1 ? 0 : (nv <= 2 ? 1 : (nv <= 4 ? 2 : 3));
  };
  const int fi = lane & 15, fq = lane >> 4;        // 16x16x32 fragment coordinates
  int gate_img = -1;                               // image whose dgate sums are in ra
  int gateL_img = -1;                              // image whose SE gate is in gateL
  // SE gate gradient sums of image `gate_img`: registers -> wgt (the row-lanes of a column: LDS atomics) -> global
  auto flush_gate = [&]() {
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int ch0 = ci * ECC + (lane & ((1 << chunk_lsh(ci)) - 1)) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (ch0 < a.KO) atomicAdd(&wgt[ch0 + e], ra[ci][e]);
        ra[ci][e] = 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < a.KO; c += 64) {
      const float v = wgt[c];
      if (v != 0.f) atomicAdd(&a.epi.dgate[(size_t)gate_img * a.KO + c], v);
      wgt[c] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (t0 < t1) issue(t0);
  for (int t = t0; t < t1; ++t) {
    const int rowS = t * srows;
    const int rows_step = min(srows, a.M - rowS);
    // ---- stage dy = a*dz + b*y + c (bf16, zero beyond R and beyond M), raw x and the old gradient
#pragma unroll
    for (int i = 0; i < NSR; ++i) {
      {
        const int r = i * a.cr.rp + rsubR;
        if (r < srows) {
          if (NOY) {         // dz is the matrix-core operand as it is (R % 8 == 0 on this path: whole chunks)
            u32x4 v = rz[i];
            if (r >= rows_step) v = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(Dt + r * a.SA + colR * 16) = v;
          } else {
            float x[8];
            unpack8(make_uint4(rz[i][0], rz[i][1], rz[i][2], rz[i][3]), x);
            if (GBN) {
              float y[8];
              const u32x4 yv = ry[(GBN && !NOY) ? i : 0];
              unpack8(make_uint4(yv[0], yv[1], yv[2], yv[3]), y);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaf(ga[e], x[e], fmaf(gb[e], y[e], gc[e]));
            }
            if (kvalid < 8 || r >= rows_step) {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (e >= kvalid || r >= rows_step) x[e] = 0.f;
            }
            *reinterpret_cast<uint4*>(Dt + r * a.SA + colR * 16) = pack8(x);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      {
        const int r = i * a.cx.rp + rsubX;
        if (r < srows) {
          *reinterpret_cast<u32x4*>(Xt + r * a.SX + colX * 16) = rx[i];
          if (has_beta) *reinterpret_cast<u32x4*>(Ot + r * a.SX + colX * 16) = ro[BETA ? i : 0];
        }
      }
    }
    if (t + 1 < t1) issue(t + 1);
    __builtin_amdgcn_wave_barrier();
    if (rows_step < srows) {
      // last, partial step: the rows past M hold copies of row M-1.  Their dy rows are zero, so they add nothing to dW,
      // to the sums or (the store is guarded) to gout -- except through NOY's Gram matrix and column sums of x: zero them
      const int slots = a.SX / 16;
      for (int q = lane; q < (srows - rows_step) * slots; q += 64)
        *reinterpret_cast<uint4*>(Xt + (size_t)(rows_step + q / slots) * a.SX + (q % slots) * 16) = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
    }

    for (int g = 0; g < a.G; ++g) {
      const int row0 = rowS + g * TR;
      if (row0 >= a.M) break;
      const int rows_valid = min(TR, a.M - row0);
      unsigned char* Dg = Dt + (size_t)g * TR * a.SA;
      unsigned char* Xg = Xt + (size_t)g * TR * a.SX;
      const unsigned char* Og = Ot + (size_t)g * TR * a.SX;
      const int img0 = row0 / a.hw;
      // STRADDLE = false: the host guarantees hw % 32 == 0 wherever a gate is involved -- a tile lies in one image
      const bool one_img = !STRADDLE || img0 == (row0 + rows_valid - 1) / a.hw;
      // SE gate of the tile's image -> LDS (once per image and wave: this load does wait behind the prefetch)
      if (gated && one_img && img0 != gateL_img) {
        for (int c = lane; c < a.KO; c += 64) gateL[c] = a.tv.gate[(size_t)img0 * a.KO + c];
        gateL_img = img0;
        __builtin_amdgcn_wave_barrier();
      }
      // dgate bookkeeping: the sums in ra belong to one image; a tile that straddles two images goes straight to
      // global atomics
      bool gate_direct = false;
      if (want_gate) {
        gate_direct = !one_img;
        if (gate_img >= 0 && (gate_direct || img0 != gate_img)) {
          flush_gate();
          gate_img = -1;
        }
        if (!gate_direct) gate_img = img0;
      }

      // ---- data gradient, 64 input channels at a time, and the activated operand for the weight gradient
      const unsigned char* arow = Dg + (size_t)j * a.SA + h * 16;
#pragma unroll
      for (int ci = 0; ci < NCH; ++ci) {
        const int c0 = ci * ECC;
        if (c0 < a.KOpad) {
          const int ccols = min(ECC, a.KOpad - c0);                // 32 or 64
          const int lsh = chunk_lsh(ci);
          const int ecol = lane & ((1 << lsh) - 1), erow = lane >> lsh, rpp = 64 >> lsh;
          const int ch0 = c0 + ecol * 8;                           // this lane's 8 channels (KO % 8 == 0: all or none)
          const bool col_ok = ch0 < a.KO;
          float sc[8], sh[8], gt[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; gt[e] = 1.f; }
          if (col_ok) {
            loadf8(coefS + ch0, sc);
            loadf8(coefH + ch0, sh);
            if (gated && one_img) loadf8(gateL + ch0, gt);
          }
          for (int nt = 0; nt < ccols / 32; ++nt) {
            f32x16 d = mma_tile(Wl + (size_t)(c0 + nt * 32 + j) * a.SW + h * 16, arow, a.ksteps);
            if (NOY)         // + x G: the rows of Xt (raw x = the forward's operand on this path) against G
              d = mma_tile_more(Gl + (size_t)(c0 + nt * 32 + j) * a.SG + h * 16, Xg + (size_t)j * a.SX + h * 16, a.kxsteps, d);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<float4*>(Ct + j * a.SC + (nt * 32 + 8 * q + 4 * h) * 4) =
                  make_float4(d[4 * q + 0], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
          }
          __builtin_amdgcn_wave_barrier();
          float vv[8];
          if (NOY) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = 0.f;
            if (col_ok) loadf8(vL + ch0, vv);
          }
          // one row of the lane's 8 channels
          auto pass = [&](int r) {
            uint4* xslot = reinterpret_cast<uint4*>(Xg + r * a.SX + (c0 / 8 + ecol) * 16);
            const float4 d0 = *reinterpret_cast<const float4*>(Ct + r * a.SC + ecol * 32);
            const float4 d1 = *reinterpret_cast<const float4*>(Ct + r * a.SC + ecol * 32 + 16);
            float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float x[8], gg[8], av[8];
            unpack8(*xslot, x);
            if (NOY) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { d[e] += vv[e]; xacc[e] += x[e]; }     // xacc: sum_p x (the s of dW)
            }
            // z (pre-activation), its activation av and the chained gradient gg
            if (swish) {
              if (want_gate) {       // the gate gradient's consumer applies act' itself: gg = d
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float z = fmaf(x[e], sc[e], sh[e]);
                  av[e] = z * sigmoidf_(z);
                  gg[e] = d[e];
                }
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float z = fmaf(x[e], sc[e], sh[e]);
                  const float sg = sigmoidf_(z);
                  av[e] = z * sg;
                  gg[e] = d[e] * (sg * (1.0f + z * (1.0f - sg)));
                }
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                av[e] = fmaf(x[e], sc[e], sh[e]);
                gg[e] = d[e];
              }
            }
            if (want_gate) {
              if (STRADDLE && gate_direct) {
                if (r < rows_valid) {
                  const int img = (row0 + r) / a.hw;
#pragma unroll
                  for (int e = 0; e < 8; ++e) atomicAdd(&a.epi.dgate[(size_t)img * a.KO + ch0 + e], d[e] * av[e]);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[ci][e] = fmaf(d[e], av[e], ra[ci][e]);
              }
            }
            if (gated) {
              if (STRADDLE && !one_img) loadf8(a.tv.gate + (size_t)(min(row0 + r, a.M - 1) / a.hw) * a.KO + ch0, gt);
#pragma unroll
              for (int e = 0; e < 8; ++e) av[e] *= gt[e];
            }
            if (has_beta) {
              float old[8];
              unpack8(*reinterpret_cast<const uint4*>(Og + r * a.SX + (c0 / 8 + ecol) * 16), old);
#pragma unroll
              for (int e = 0; e < 8; ++e) gg[e] += old[e];
            }
            if (r < rows_valid) *reinterpret_cast<uint4*>(GO + (size_t)(row0 + r) * a.tv.ld + ch0) = pack8(gg);
            *xslot = pack8(av);
            if (want_stats) {      // raw sums (g, g*x): sum g*(x-mean)*rstd is taken from the totals at the end
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                ra[ci][e] += gg[e];
                rb[ci][e] = fmaf(gg[e], x[e], rb[ci][e]);
              }
            }
          };
          if (col_ok) {
            if (lsh == 3) {
#pragma unroll
              for (int p = 0; p < 4; ++p) pass(p * 8 + erow);
            } else {
              for (int r = erow; r < TR; r += rpp) pass(r);
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }

      // ---- weight gradient: dW[k][n] += sum over the tile's 32 rows of xa[row][k] * dy[row][n]
      {
        const unsigned char* St = ksmall ? Xg : Dg;      // small side: fragments kept in registers
        const unsigned char* Lt = ksmall ? Dg : Xg;
        const int sS = ksmall ? a.SX : a.SA, sL = ksmall ? a.SA : a.SX;
        bf16x8 sf[FT_S];
#pragma unroll
        for (int s = 0; s < FT_S; ++s)
          if (s < TS) sf[s] = column_frag_tr(St, sS, fq * 8, s * 16, fi);
        if (NOY) {           // S = x^T x (ksmall on this path: sf are the fragments of x)
#pragma unroll
          for (int s = 0; s < FT_S; ++s)
#pragma unroll
            for (int s2 = 0; s2 < FT_S; ++s2)
              if (s < TS && s2 < TS)
                accS[NOY ? s : 0][NOY ? s2 : 0] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[s], sf[s2], accS[NOY ? s : 0][NOY ? s2 : 0], 0, 0, 0);
        }
#pragma unroll
        for (int l = 0; l < FT_L; ++l) {
          if (l < TL) {
            const bf16x8 lf = column_frag_tr(Lt, sL, fq * 8, l * 16, fi);
#pragma unroll
            for (int s = 0; s < FT_S; ++s) {
              if (s < TS) {
                // A = the k side (rows of dW), B = the n side (columns of dW)
                if (ksmall) acc[s][l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[s], lf, acc[s][l], 0, 0, 0);
                else acc[s][l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, sf[s], acc[s][l], 0, 0, 0);
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (want_gate && gate_img >= 0) flush_gate();
  if (NOY) {     // the lanes' column sums -> wst[0 .. KO) (the row-lanes of a column: LDS atomics, once per kernel)
    const int ch0 = (lane & ((1 << chunk_lsh(0)) - 1)) * 8;
    if (ch0 < a.KO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&wst[ch0 + e], xacc[e]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float xsum = (NOY && lane < a.KO) ? wst[lane] : 0.f;      // NOY: this wave's sum_p x[p][lane] (KO <= 32)
  if (want_stats) {
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int ch0 = ci * ECC + (lane & ((1 << chunk_lsh(ci)) - 1)) * 8;
      if (ch0 < a.KO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          atomicAdd(&wst[ch0 + e], ra[ci][e]);
          atomicAdd(&wst[a.KOpad + ch0 + e], rb[ci][e]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // the four waves in wave order (r04: no cross-wave LDS atomics -- the same sums on every run)
    for (int w = 0; w < WAVES; ++w) {
      if (wave == w) {
        for (int c = lane; c < a.KO; c += 64) {
          red[c] += wst[c];
          red[a.KOpad + c] += wst[a.KOpad + c];
        }
      }
      __syncthreads();
    }
    float* dst = a.epi.stat_partials + (size_t)blockIdx.x * 2 * a.KO;
    for (int c = tid; c < a.KO; c += THREADS) {
      const float sg = red[c], sgx = red[a.KOpad + c];
      dst[c] = sg;
      dst[a.KO + c] = a.epi.rstd[c] * (sgx - a.epi.mean[c] * sg);      // sum g*(x-mean)*rstd
    }
  }
  // ---- dW blocks of the four waves -> one partial per workgroup (fixed order: deterministic)
  __syncthreads();
  float* scratch = reinterpret_cast<float*>(smem);      // [KO][R] fp32; every LDS tile is dead by now
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int s = 0; s < FT_S; ++s)
#pragma unroll
        for (int l = 0; l < FT_L; ++l) {
          if (s < TS && l < TL) {
            const int tk = ksmall ? s : l, tn = ksmall ? l : s;
            const int n = tn * 16 + fi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = tk * 16 + fq * 4 + e;
              if (k < a.KO && n < a.R) {
                float* p = scratch + (size_t)k * a.R + n;
                *p = (w == 0 ? 0.f : *p) + acc[s][l][e];
              }
            }
          }
        }
    }
    __syncthreads();
  }
  const size_t part = (size_t)a.KO * a.R + (NOY ? (size_t)a.KO * a.KO + a.KO : 0);
  float* dstw = a.ws + (size_t)blockIdx.x * part;
  for (int i = tid; i < a.KO * a.R; i += THREADS) dstw[i] = scratch[i];
  if (NOY) {
    // [S | s] behind P: the waves' Gram tiles and column sums in wave order (deterministic)
    __syncthreads();
    float* sS = scratch;                                  // [KO][KO]
    float* ss = scratch + (size_t)a.KO * a.KO;            // [KO]
    for (int w = 0; w < WAVES; ++w) {
      if (wave == w) {
#pragma unroll
        for (int s = 0; s < FT_S; ++s)
#pragma unroll
          for (int s2 = 0; s2 < FT_S; ++s2) {
            if (s < TS && s2 < TS) {
              const int k2 = s2 * 16 + fi;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int k = s * 16 + fq * 4 + e;
                if (k < a.KO && k2 < a.KO) {
                  float* p = sS + (size_t)k * a.KO + k2;
                  *p = (w == 0 ? 0.f : *p) + accS[NOY ? s : 0][NOY ? s2 : 0][e];
                }
              }
            }
          }
        if (lane < a.KO) ss[lane] = (w == 0 ? 0.f : ss[lane]) + xsum;
      }
      __syncthreads();
    }
    for (int i = tid; i < a.KO * a.KO + a.KO; i += THREADS) dstw[(size_t)a.KO * a.R + i] = scratch[i];
  }
}

// NOY: dW[k][n] += a[n] P[k][n] + b[n] sum_k' S[k][k'] W[k'][n] + c[n] s[k] from the summed partial [P | S | s]
__global__ __launch_bounds__(256) void k_noy_apply(const float* __restrict__ psum, const bf16_t* __restrict__ W, int ldw,
                                                  int KO, int R, const float* __restrict__ ga,
                                                  const float* __restrict__ gb, const float* __restrict__ gc,
                                                  float* __restrict__ dweight) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= KO * R) return;
  const int k = i / R, n = i - k * R;
  const float* S = psum + (size_t)KO * R + (size_t)k * KO;
  float t = 0.f;
  for (int k2 = 0; k2 < KO; ++k2) t = fmaf(S[k2], bf2f(W[(size_t)k2 * ldw + n]), t);
  dweight[i] += ga[n] * psum[i] + gb[n] * t + gc[n] * psum[(size_t)KO * R + (size_t)KO * KO + k];
}

// lab switches (read per call): integer environment variable or the default
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

template <typename KernelT>
inline bool allow_big_lds(KernelT kern, size_t lds) {
  if (lds <= 64 * 1024) return true;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024) == hipSuccess;
}

}  // namespace pws

// Returns 1 when the streaming kernel handled the call, 0 when the shape is outside its envelope
// (the caller then falls back to the tiled kernel in pw_gemm.hip), negative on error.
int pws_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, hipStream_t st) {
  using namespace pws;
  const int K = in->c, N = cout;
  if (K > 256 || K % 8 != 0) return 0;
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Wt = reinterpret_cast<const bf16_t*>(wt); a.ldw = ldw; a.bias = bias;
  a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo;
  a.M = in->n * in->h * in->w; a.K = K; a.N = N; a.hw = in->h * in->w;
  a.stat_partials = stat_partials;
  a.ck = make_colmap(K);
  const bool oact_ = in->act > EDET_ACT_SWISH;
  int NS = K > 128 ? 16 : 8;
  a.ksteps = (K + 15) / 16;
  a.SA = frag_stride(K, K % 16 != 0);
  a.SW = a.SA;
  a.Npad = (N + 31) / 32 * 32;
  a.nvec_out = (N + 7) / 8;
  if (ldo < a.nvec_out * 8) return 0;
  // rows kept in flight per wave: as many 32-row tiles as NS load passes cover, shrunk until three
  // workgroups fit in a CU's LDS (53 KB each); the weight chunk may take it to one workgroup per CU
  size_t lds = 0;
  for (a.G = NS * a.ck.rp / TR > 4 ? 4 : NS * a.ck.rp / TR; ; --a.G) {
    if (a.G < 1) a.G = 1;
    a.pst = (TR * a.G + a.ck.rp - 1) / a.ck.rp;
    if (a.pst > NS) return 0;
    lds = plan_lds(a.Npad, a.pst * a.ck.rp, a.SA, a.SW, 53 * 1024, &a.NWC, &a.SC);
    if ((lds != 0 && a.NWC == a.Npad) || a.G == 1) break;
  }
  if (lds == 0 || a.NWC != a.Npad) {
    lds = plan_lds(a.Npad, a.pst * a.ck.rp, a.SA, a.SW, 80 * 1024, &a.NWC, &a.SC);
    if (lds == 0 || a.NWC != a.Npad) lds = plan_lds(a.Npad, a.pst * a.ck.rp, a.SA, a.SW, 150 * 1024, &a.NWC, &a.SC);
    if (lds == 0) return 0;
  }
  if (!oact_ && env_int("EDET_PWS_FWD_EXACT", 1)) {
    // the instantiation with the fewest passes that covers the super-tile; its passes are all issued (rows beyond the
    // super-tile are the wave's next rows) and staged, so the LDS tile holds NS * rp rows
    static const int exact_ns[] = {4, 6, 7, 8, 11, 16};
    int pick = 0;
    for (int q = 0; q < 6; ++q) if (exact_ns[q] >= a.pst) { pick = exact_ns[q]; break; }
    if (pick > 0) {
      int nwc2 = 0, sc2 = 0;
      const size_t cap = lds <= 53 * 1024 ? 53 * 1024 : (lds <= 80 * 1024 ? 80 * 1024 : 150 * 1024);
      const size_t l2 = plan_lds(a.Npad, pick * a.ck.rp, a.SA, a.SW, cap, &nwc2, &sc2);
      if (l2 != 0 && nwc2 == a.NWC) { NS = -pick; a.pst = pick; lds = l2; a.SC = sc2; }
    }
  }
  a.nwc = (a.Npad + a.NWC - 1) / a.NWC;
  if (a.nwc > 2) return 0;   // the big operand would be re-read too often: leave it to the tiled kernel
  const int nst = (a.M + TR * a.G - 1) / (TR * a.G);
  // >= 4 super-tiles per wave (r03d lab, 24 pointwise layer shapes: 2 / 4 / 8 / 16 -> forward 4.85 / 4.79 / 4.82 / 4.87 ms)
  const int spw_min = env_int("EDET_PWS_SPW", 4);
  int grid = (nst + WAVES * spw_min - 1) / (WAVES * spw_min);
  // One round: at most as many workgroups as the chip holds at once (3 per compute unit for the 8-pass kernel, 2 for the
  // 16-pass one).  r03h lab, caps of 1024 (the partial-row limit, round 2) / 768 / 512: 320x320x16->96 0.92 / 0.81 / 0.83
  // ms, 160x160x24->144 0.39 / 0.33 / 0.36, 320x320x32->16 0.46 / 0.40 / 0.48, 80x80x64->64 62 / 55 / 60 us; the
  // 144-channel inputs (2 per CU) 0.36 / 0.38 / 0.36: with 1024 workgroups on 768 slots the second round runs a third full.
  const void* kfn = nullptr;
  switch (NS) {       // negative: the EXACT instantiation with -NS passes
    case -4: kfn = reinterpret_cast<const void*>(&k_pw_fwd<4, true, false>); break;
    case -6: kfn = reinterpret_cast<const void*>(&k_pw_fwd<6, true, false>); break;
    case -7: kfn = reinterpret_cast<const void*>(&k_pw_fwd<7, true, false>); break;
    case -8: kfn = reinterpret_cast<const void*>(&k_pw_fwd<8, true, false>); break;
    case -11: kfn = reinterpret_cast<const void*>(&k_pw_fwd<11, true, false>); break;
    case -16: kfn = reinterpret_cast<const void*>(&k_pw_fwd<16, true, false>); break;
    case 8: kfn = oact_ ? reinterpret_cast<const void*>(&k_pw_fwd<8, false, true>) : reinterpret_cast<const void*>(&k_pw_fwd<8, false, false>); break;
    default: kfn = oact_ ? reinterpret_cast<const void*>(&k_pw_fwd<16, false, true>) : reinterpret_cast<const void*>(&k_pw_fwd<16, false, false>); break;
  }
  const int slots_fwd = edet_resident_wgs(kfn, THREADS, lds);
  const int cap_fwd = env_int("EDET_PWS_FWD_CAP", slots_fwd > 0 ? slots_fwd : EDET_MAX_PARTS);     // lab switch overrides
  if (grid > cap_fwd) grid = cap_fwd;
  if (grid > EDET_MAX_PARTS) grid = EDET_MAX_PARTS;
  if (grid < 1) grid = 1;
  a.spw = (nst + grid * WAVES - 1) / (grid * WAVES);
  grid = (nst + a.spw * WAVES - 1) / (a.spw * WAVES);
  if (nparts_out) *nparts_out = grid;
#define PWS_FWD(NS_, OACT_, EXACT_)                                                      \
  do {                                                                                  \
    if (!allow_big_lds(&k_pw_fwd<NS_, EXACT_, OACT_>, lds)) return 0;                   \
    edet_launch(k_pw_fwd<NS_, EXACT_, OACT_>, dim3(grid), dim3(THREADS), lds, st, a);   \
  } while (0)
  switch (NS) {
    case -4: PWS_FWD(4, false, true); break;
    case -6: PWS_FWD(6, false, true); break;
    case -7: PWS_FWD(7, false, true); break;
    case -8: PWS_FWD(8, false, true); break;
    case -11: PWS_FWD(11, false, true); break;
    case -16: PWS_FWD(16, false, true); break;
    case 8: if (oact_) PWS_FWD(8, true, false); else PWS_FWD(8, false, false); break;
    default: if (oact_) PWS_FWD(16, true, false); else PWS_FWD(16, false, false); break;
  }
#undef PWS_FWD
  EDET_LAUNCH_CHECK("edet_pw_fwd(stream)");
  return 1;
}

int pws_try_dgrad(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st) {
  using namespace pws;
  const int R = dy->c, KO = in->c;
  if (R > 128 || KO > 512 || KO % 8 != 0 || dy->ld % 8 != 0) return 0;
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gv = *dy; a.tv = *in; a.W = reinterpret_cast<const bf16_t*>(w); a.ldw = ldw; a.epi = *epi;
  a.M = in->n * in->h * in->w; a.R = R; a.KO = KO; a.hw = in->h * in->w;
  a.cr = make_colmap(R);
  const bool gbn = dy->a != nullptr;
  const int NS = 8;
  a.ksteps = (R + 15) / 16;
  a.SA = frag_stride((R + 7) / 8 * 8, ((R + 7) / 8 * 8) % 16 != 0);
  a.SW = a.SA;
  a.KOpad = (KO + 31) / 32 * 32;
  a.SC = ECC * 4 + 16;
  a.nvec_out = KO / 8;
  size_t lds = 0;
  int g0 = NS * a.cr.rp / TR;
  if (gbn) g0 /= 2;                    // two tensors per row: half the rows for the same bytes in flight
  if (g0 > 4) g0 = 4;
  for (a.G = g0;; --a.G) {
    if (a.G < 1) a.G = 1;
    a.pst = (TR * a.G + a.cr.rp - 1) / a.cr.rp;
    if (a.pst > NS) return 0;
    lds = (size_t)a.KOpad * a.SW + (size_t)2 * a.KOpad * 4 +
          (size_t)WAVES * ((size_t)a.pst * a.cr.rp * a.SA + (size_t)TR * a.SC + (size_t)3 * a.KOpad * 4);
    if (lds <= 53 * 1024 || a.G == 1) break;
  }
  if (lds > 150 * 1024) return 0;
  const int nst = (a.M + TR * a.G - 1) / (TR * a.G);
  const int spw_min = env_int("EDET_PWS_SPW", 4);      // r03d lab: 2 / 4 / 8 -> backward 16.62 / 16.49 / 16.45 ms over 24 shapes
  int grid = (nst + WAVES * spw_min - 1) / (WAVES * spw_min);
  const int cap_bwd = env_int("EDET_PWS_BWD_CAP", EDET_MAX_PARTS);     // lab switch
  if (grid > cap_bwd) grid = cap_bwd;
  if (grid > EDET_MAX_PARTS) grid = EDET_MAX_PARTS;
  if (grid < 1) grid = 1;
  a.spw = (nst + grid * WAVES - 1) / (grid * WAVES);
  grid = (nst + a.spw * WAVES - 1) / (a.spw * WAVES);
  if (nparts_out) *nparts_out = grid;
#define PWS_DGRAD(NS_, GBN_, OACT_)                                                    \
  do {                                                                                 \
    if (!allow_big_lds(&k_pw_dgrad<NS_, GBN_, OACT_>, lds)) return 0;                  \
    edet_launch(k_pw_dgrad<NS_, GBN_, OACT_>, dim3(grid), dim3(THREADS), lds, st, a);  \
  } while (0)
  if (in->act > EDET_ACT_SWISH) { if (gbn) PWS_DGRAD(8, true, true); else PWS_DGRAD(8, false, true); }
  else { if (gbn) PWS_DGRAD(8, true, false); else PWS_DGRAD(8, false, false); }
#undef PWS_DGRAD
  EDET_LAUNCH_CHECK("edet_pw_bwd_data(stream)");
  return 1;
}

int pws_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st) {
  using namespace pws;
  const int K = in->c, N = dy->c;
  if (!workspace || K % 8 != 0 || dy->ld % 8 != 0 || in->ld % 8 != 0) return 0;
  const bool ug = N >= K;
  WgArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy; a.ws = reinterpret_cast<float*>(workspace);
  a.M = in->n * in->h * in->w; a.K = K; a.N = N; a.hw = in->h * in->w;
  a.CU = ug ? N : K; a.CV = ug ? K : N;
  if (a.CV > 64) return 0;
  const bool gbn = dy->a != nullptr;
  if (gbn && N % 8 != 0) return 0;
  a.nus = (a.CU + 63) / 64;
  const int nvecV = (a.CV + 7) / 8;
  a.cpwV = 1;
  while (a.cpwV < nvecV) a.cpwV <<= 1;
  // row splits: ~2048 waves in total, at least 8 steps of 32 rows each, bounded by the workspace
  const int64_t kn = (int64_t)K * N;
  int S = env_int("EDET_PWS_WG_TARGET", 2048) / a.nus;
  const int max_by_rows = (a.M + 8 * TR - 1) / (8 * TR);
  if (S > max_by_rows) S = max_by_rows;
  const int64_t max_by_ws = (int64_t)(workspace_bytes / sizeof(float)) / kn;
  if (S > max_by_ws) S = (int)max_by_ws;
  if (S < 1) return 0;
  a.rows_per_split = ((a.M + S - 1) / S + TR - 1) / TR * TR;
  a.S = (a.M + a.rows_per_split - 1) / a.rows_per_split;
  int grid;
  if (a.nus >= 3) grid = ((a.nus + 3) / 4) * a.S;
  else grid = (a.S + WAVES / a.nus - 1) / (WAVES / a.nus);
#define PWS_WG(UG_, GBN_)                                                                          \
  do {                                                                                             \
    if (in->act > EDET_ACT_SWISH) edet_launch(k_pw_wgrad<UG_, GBN_, true>, dim3(grid), dim3(THREADS), 0, st, a); \
    else edet_launch(k_pw_wgrad<UG_, GBN_, false>, dim3(grid), dim3(THREADS), 0, st, a);           \
  } while (0)
  if (ug) { if (gbn) PWS_WG(true, true); else PWS_WG(true, false); }
  else { if (gbn) PWS_WG(false, true); else PWS_WG(false, false); }
#undef PWS_WG
  EDET_LAUNCH_CHECK("edet_pw_bwd_weight(stream)");
  if (edet_reduce_partials(a.ws, a.S, kn, dweight, st) != 0) return -2;
  return 1;
}

// Both gradients of one pointwise layer in one pass: 1 = handled, 0 = outside the envelope (the caller runs the two
// separate kernels), < 0 = error.  dweight [KO][R] fp32 is accumulated into (edet_reduce_partials).
int pws_try_bwd_fused(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                      const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  if (in->act > EDET_ACT_SWISH) return 0;     // relu / relu6 / hswish: the two streaming kernels (OACT instantiations)
  using namespace pws;
  const int R = dy->c, KO = in->c;
  if (!workspace || KO % 8 != 0 || dy->ld % 8 != 0 || in->ld % 8 != 0 || R > 160 || KO > 160) return 0;
  if (epi->stat_partials && epi->dgate) return 0;     // one set of running sums per lane: BatchNorm backward OR gate
  if (in->gate && epi->dgate) return 0;               // r06: this kernel's gate-gradient sums are atomics -- the tiled one-pass kernel
                                                      // (ordered slots) or the two-kernel path + k_gate_sums take SE-gated inputs
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.gv = *dy; a.tv = *in; a.W = reinterpret_cast<const bf16_t*>(w); a.ldw = ldw; a.epi = *epi;
  a.ws = reinterpret_cast<float*>(workspace);
  a.M = in->n * in->h * in->w; a.R = R; a.KO = KO; a.hw = in->h * in->w;
  a.TK = (KO + 15) / 16; a.TN = (R + 15) / 16;
  // Envelope = the instantiations below.  Class A, the "expand" shape (few input channels, many output channels:
  // R >= 2 KO, dW <= 2 x 9 tiles of 16 x 16).  r02d, D0 640x640 batch 128, both gradients: 320x320x16->96 2.44 -> 1.95
  // ms, 160x160x24->144 1.45 -> 0.73 ms.  Class B (r03): at most 64 output channels -- the project layers (dW <= 3 x 9
  // tiles) and the 64 -> 64 layers (4 x 4) -- which the round-2 kernel ran slower than the two-kernel path because it
  // issued 5-8x the gradient loads a tile needs (every instantiated pass) and paid 16-48 LDS atomics per tile for the
  // per-channel sums; EDET_PWS_FUSED_WIDE=0 keeps class A only (lab switch, read per call).
  const int tmin = a.TK < a.TN ? a.TK : a.TN, tmax = a.TK < a.TN ? a.TN : a.TK;
  const bool class_a = R >= 2 * KO && a.TK <= 2 && a.TN <= 9;
  int ft = 0;                                           // class B tile grid: 44 = 4 x 4, 39 = 3 x 9
  if (!class_a) {
    if (env_int("EDET_PWS_FUSED_WIDE", 1) == 0) return 0;
    // project shape only.  r03m: the 64 -> 64 layers of the BiFPN / heads (plain input view, no chain epilogue) take
    // 0.181 ms one-pass against 0.169 ms for the two tiled kernels at 80x80, and a kernel that owns every register of
    // a CU shuts out the small pyramid levels' chain on the second stream (the step did not move: 64.6 ms)
    if (KO < 2 * R) return 0;
    if (R > 64 || epi->beta) return 0;       // class B instantiations do not accumulate into gout
    // r04: the one-pass TILED kernel (pw_tile_bwd.hip) takes the project layers from 96 input channels up -- lab, D0
    // 640x640 batch 128: 160x160x96->24 0.436 against 0.582 ms here, 80x80x144->40 0.226 / 0.274, 160x160x144->24 0.803 /
    // 0.796 (a tie, and the tiled kernel has no atomics); 320x320x32->16 stays here (0.787 against 0.841 ms).
    // EDET_PWT=0 (the tiled kernel off) restores the round-3 envelope.
    {
      const char* pwt_env = getenv("EDET_PWT");
      // ... and every SE-gated projection: this kernel adds its gate-gradient sums into dgate with global atomics, the
      // tiled kernel in a fixed order (320x320x32->16: 0.84 against 0.79 ms here -- the price of a reproducible step)
      if ((KO > 32 || in->gate) && !(pwt_env && pwt_env[0] == '0')) return 0;
    }
    if (in->gate && a.hw % TR != 0) return 0;   // ... and take every 32-row tile to lie in one image where a gate is involved
    if (tmax <= 4 && KO <= 64) ft = 44;
    else if (tmin <= 3 && tmax <= 9) ft = 39;
    else return 0;
    // r03k lab, 64 -> 64: 80x80 (819 K rows) 0.219 -> 0.165 ms, 40x40 (205 K rows) 0.071 -> 0.070, 20x20 0.030 -> 0.054:
    // the prologue, the LDS reduction of dW and one wave per SIMD need ~250 K rows to pay off
    if (a.M < env_int("EDET_PWS_FUSED_MINROWS", 262144)) return 0;
  }
  a.cr = make_colmap(R);
  a.cx = make_colmap(KO);
  const int Rp = (R + 7) / 8 * 8;
  a.ksteps = (R + 15) / 16;
  a.SA = frag_stride(Rp, Rp % 16 != 0);
  a.SW = a.SA;
  a.SX = frag_stride(KO, false);
  a.KOpad = (KO + 31) / 32 * 32;
  if (a.SX < a.KOpad * 2) a.SX = frag_stride(a.KOpad, false);     // the epilogue addresses whole 32-channel tiles
  a.SC = ECC * 4 + 16;
  const bool gbn = dy->a != nullptr;
  // NOY: BatchNorm backward without the saved convolution output (see the kernel's header): dy carries a BatchNorm
  // backward, the input is a plain stored tensor (x~ = x; no chain epilogue), whole 16-byte chunks of dz.
  // EDET_PW_NOY=0 keeps the form that reads y (lab switch, read per call).  r03c, first version (column sums of x through
  // 16 LDS atomics per tile): SLOWER than reading y, 320x320x16->96 1.86 -> 2.31 ms -- the kernel runs one wave per SIMD
  // with one tile of loads in flight and is bound by that latency, not by bytes.  r03d, sums kept in registers across the
  // tiles: 1.90 -> 1.75 ms, 160x160x24->144 0.76 -> 0.68 ms (-8 / -10 %), with 43 % less traffic.
  const char* noy_env = getenv("EDET_PW_NOY");
  const bool noy = class_a && gbn && (epi->flags & EDET_EPI_Y_IS_CONV_OF_INPUT) && dy->b && dy->cc && !in->scale && !in->gate &&
                   in->act == EDET_ACT_NONE &&
                   !epi->stat_partials && !epi->dgate && R % 8 == 0 && KO <= 32 && !(noy_env && noy_env[0] == '0');
  a.SG = frag_stride(a.KOpad, false);
  a.kxsteps = (KO + 15) / 16;
  const size_t part = (size_t)KO * R + (noy ? (size_t)KO * KO + KO : 0);      // floats per workgroup partial
  // Rows in flight: G tiles of 32 rows per step, as many as an instantiated (NSR, NSX) pair of load passes covers
  // without loading more than ~30 % past the step, the LDS (150 KB) and a budget of ~24 KB of loads per wave allow (the
  // kernel runs one wave per SIMD: the bytes in flight per wave ARE the latency hiding).  EDET_PWS_FUSED_G caps it
  // (lab switch).  r03j, D0 640x640 batch 128: 320x320x32->16 1.87 / 1.57 / 1.43 ms at G = 1 / 2 / 4.
  static const int pairs_noy[][2] = {{13, 2}, {22, 4}};
  static const int pairs_a[][2] = {{12, 4}};
  static const int pairs_44[][2] = {{4, 8}};
  static const int pairs_39[][2] = {{2, 11}, {3, 11}, {2, 7}};
  const int (*pairs)[2] = class_a ? (noy ? pairs_noy : pairs_a) : (ft == 44 ? pairs_44 : pairs_39);
  const int npairs = class_a ? (noy ? 2 : 1) : (ft == 44 ? 1 : 3);
  const int rbytes = ((gbn && !noy) ? 2 : 1) * Rp * 2, xbytes = (epi->beta ? 2 : 1) * KO * 2;
  const int g_cap = env_int("EDET_PWS_FUSED_G", 4);
  const int inflight = env_int("EDET_PWS_FUSED_INFLIGHT", 24 * 1024);
  size_t lds = 0;
  int pick = -1;
  double pick_waste = 1e30;
  a.G = 0;
  for (int g = g_cap < 4 ? (g_cap < 1 ? 1 : g_cap) : 4; g >= 1; --g) {
    const int nsr = (TR * g + a.cr.rp - 1) / a.cr.rp, nsx = (TR * g + a.cx.rp - 1) / a.cx.rp;
    if (g > 1 && TR * g * (rbytes + xbytes) > inflight) continue;
    const size_t l = (size_t)a.KOpad * a.SW + (size_t)4 * a.KOpad * 4 +
                     (noy ? (size_t)a.KOpad * 4 + (size_t)a.KOpad * a.SG : 0) +
                     (size_t)WAVES * ((size_t)TR * g * a.SA + (size_t)(epi->beta ? 2 : 1) * TR * g * a.SX +
                                      (size_t)TR * a.SC + (size_t)4 * a.KOpad * 4);
    if (l > 150 * 1024) continue;
    for (int q = 0; q < npairs; ++q) {
      if (pairs[q][0] < nsr || pairs[q][1] < nsx) continue;
      const double waste = ((double)(pairs[q][0] * a.cr.rp - TR * g) * rbytes + (double)(pairs[q][1] * a.cx.rp - TR * g) * xbytes) /
                           ((double)TR * g * (rbytes + xbytes));
      if (waste < pick_waste - 1e-9 && (pick < 0 || pick_waste > 0.3)) {
        pick = q; pick_waste = waste; a.G = g; lds = l;
      }
    }
    if (pick >= 0 && pick_waste <= 0.3) break;
  }
  if (pick < 0 || lds < (size_t)KO * R * 4 + (noy ? (size_t)(KO * KO + KO) * 4 : 0)) return 0;
  const int nstep = (a.M + TR * a.G - 1) / (TR * a.G);
  // one workgroup per compute unit, a single round (r03d lab: grids of 1024 / 512 / 256 -> 320x320x16->96 1.90 / 1.83 /
  // 1.81 ms, 160x160x24->144 0.69 / 0.64 / 0.61 ms: every workgroup pays the prologue and the dW partial once); at
  // least 2 steps per wave, partials bounded by the workspace
  int grid = env_int("EDET_PWS_FUSED_GRID", 256);
  const int64_t max_by_ws = (int64_t)(workspace_bytes / sizeof(float)) / (int64_t)part - (noy ? 1 : 0);
  if (grid > max_by_ws) grid = (int)max_by_ws;
  if (grid > EDET_MAX_PARTS) grid = EDET_MAX_PARTS;
  const int max_by_steps = (nstep + WAVES * 2 - 1) / (WAVES * 2);
  if (grid > max_by_steps) grid = max_by_steps;
  if (grid < 1) return 0;
  a.tpw = (nstep + grid * WAVES - 1) / (grid * WAVES);
  grid = (nstep + a.tpw * WAVES - 1) / (a.tpw * WAVES);
  if (nparts_out) *nparts_out = grid;
#define PWS_FUSED(NSR_, NSX_, GBN_, FS_, FL_, NOY_, NCH_, BETA_)                                                       \
  do {                                                                                                                 \
    if (!allow_big_lds(&k_pw_bwd_fused<NSR_, NSX_, GBN_, FS_, FL_, NOY_, NCH_, BETA_, BETA_>, lds)) return 0;                 \
    edet_launch(k_pw_bwd_fused<NSR_, NSX_, GBN_, FS_, FL_, NOY_, NCH_, BETA_, BETA_>, dim3(grid), dim3(THREADS), lds, st, a); \
  } while (0)
#define PWS_FUSED_GB(NSR_, NSX_, FS_, FL_, NCH_, BETA_)                    \
  do {                                                                     \
    if (gbn) PWS_FUSED(NSR_, NSX_, true, FS_, FL_, false, NCH_, BETA_);    \
    else PWS_FUSED(NSR_, NSX_, false, FS_, FL_, false, NCH_, BETA_);       \
  } while (0)
  if (class_a) {
    if (noy) {
      if (pick == 0) PWS_FUSED(13, 2, true, 2, 9, true, 1, true);
      else PWS_FUSED(22, 4, true, 2, 9, true, 1, true);
    } else {
      PWS_FUSED_GB(12, 4, 2, 9, 1, true);
    }
  } else if (ft == 44) {
    PWS_FUSED_GB(4, 8, 4, 4, 1, false);
  } else {
    if (pick == 0) PWS_FUSED_GB(2, 11, 3, 9, 3, false);
    else if (pick == 1) PWS_FUSED_GB(3, 11, 3, 9, 3, false);
    else PWS_FUSED_GB(2, 7, 3, 9, 3, false);
  }
#undef PWS_FUSED_GB
#undef PWS_FUSED
  EDET_LAUNCH_CHECK("edet_pw_bwd(fused)");
  if (noy) {
    // partials [grid][P | S | s] -> their sum behind them in the workspace -> dW
    float* psum = a.ws + (size_t)grid * part;
    if (edet_reduce_partials_set(a.ws, grid, (int64_t)part, psum, st) != 0) return -2;
    edet_launch(k_noy_apply, dim3((KO * R + 255) / 256), dim3(256), 0, st, psum, a.W, ldw, KO, R, dy->a, dy->b, dy->cc, dweight);
    EDET_LAUNCH_CHECK("edet_pw_bwd(noy apply)");
    return 1;
  }
  if (edet_reduce_partials(a.ws, grid, (int64_t)KO * R, dweight, st) != 0) return -2;
  return 1;
}
