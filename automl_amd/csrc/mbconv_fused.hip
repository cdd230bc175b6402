// Head of an MBConv block in ONE kernel for the bf16 path on gfx950: expansion 1x1 convolution -> BatchNorm ->
// activation -> depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF 'SAME').
//
// Reference: efficientdet/backbone/efficientnet_model.py:378-392 (MBConvBlock.call: x = act(bn0(expand_conv(x)));
// x = act(bn1(depthwise_conv(x)))), the layers built at :304-327.  The expanded tensor is the widest tensor of the
// block (6 x the block input): the two-kernel path (edet_pw_fwd, edet_dw_fwd) writes it once and reads it back once.
// Here the depthwise march (dw_march.hip, k_fwd_v2) gets the rows of the expanded tensor from the matrix cores instead
// of from HBM: a workgroup owns 48 expanded channels (three 16-row MFMA blocks; every EfficientNet expansion width is
// a multiple of 48) and a window of 64 input columns (4 waves x 16 pixels), marches down the image, and per input row
//   * every lane loads ONE 16-byte chunk of the block input (8 of the <= 32 input channels of its pixel -- the B
//     operand of v_mfma_f32_16x16x32_bf16, straight from global memory, prefetched a few rows ahead), applies the
//     producer's BatchNorm on load, and the wave multiplies it with three resident 16 x 32 slices of the expansion
//     kernel: D = W^T X^T, so a lane receives 4 consecutive channels x 3 blocks of ITS pixel;
//   * the 12 values are rounded to bf16 (what the two-kernel path stores and reads back), optionally stored (training:
//     the backward pass reads the raw expanded tensor), sent through the expansion BatchNorm + activation and parked
//     in a two-row fp32 LDS ring;
//   * after one barrier the same threads, now as (output pixel, channel chunk), read the K neighbours of the row back
//     and accumulate the depthwise taps into ceil(K / S) output rows held in registers, exactly as k_fwd_v2 does.
// The window overlaps its right neighbour by K - S columns (and a row tile its lower neighbour by K - S rows): 1.5-6 %
// of the expansion is computed twice, nothing is exchanged between workgroups.
//
// In training the statistics of the expansion's BatchNorm must exist before the depthwise convolution can run:
// k_exp_stats makes them in a first pass over the block input only (same MFMA, same rounding, nothing stored);
// inference needs no such pass and never stores the expanded tensor at all.
#include <stdlib.h>

#include "common.h"

namespace mbf {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t brsrc_t;

constexpr int THREADS = 256; // k_exp_stats
constexpr int GC = 48;      // expanded channels per wave: three 16-row MFMA blocks
// Every global store of the march is issued on every step, by every lane: what must not be written goes to this offset,
// beyond the descriptor's num_records (2^31 - 1), where the hardware drops it.  A store under a branch -- even a uniform
// one -- makes the compiler's count of the memory operations in flight unknowable, and every wait on the block-input
// FIFO then becomes a wait for (nearly) ALL of them, the stores of the expanded row included (r06 lab: s_waitcnt vmcnt(1)
// in front of every store and MFMA; the training kernel took the inference kernel's time PLUS the time of its stores).
constexpr uint32_t OOB = 0x80000000u;

__device__ __forceinline__ brsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__host__ __device__ constexpr int gcd_(int x, int y) { return y == 0 ? x : gcd_(y, x % y); }
__host__ __device__ constexpr int fdiv_(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__host__ __device__ constexpr int slot_of(int rel, int n) { return ((rel % n) + n) % n; }
struct URange {
  int lo; uint32_t span;
  __device__ __forceinline__ void set(int l, int h) { lo = l; span = h > l ? (uint32_t)(h - l) : 0u; }
  __device__ __forceinline__ bool has(int v) const { return (uint32_t)(v - lo) < span; }
};

struct Args {
  edet_tview_t in;            // block input [n][H][W][cin]: affine view (the producer's BatchNorm) or a stored tensor
  const bf16_t* wt; int ldw;  // expansion kernel, transposed: [cexp][ldw], input channel contiguous
  int cexp;
  const float* esc; const float* esh; int act;   // expansion BatchNorm (scale, shift), activation code
  bf16_t* e_out; int lde;     // raw expanded tensor (training) or NULL
  const float* dww;           // depthwise kernel [K][K][cexp] fp32
  bf16_t* out; int ldo;
  float* stat_partials;       // [P][2][cexp] of the depthwise output (training) or NULL
  int oh, ow, pad_t, pad_l;
  int ngroups, TY, tiles_x, tiles_y, ntiles, P;
  int ngb;                    // channel blocks (workgroups) per tile slot: cexp / (48 NGR)
  long long M;                // k_exp_stats: pixels
  int dbg;                    // lab switch EDET_MBF_DBG (1: the stores of the expanded tensor are dropped)
};

__device__ __forceinline__ bf16x8 zero_frag() { return __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u)); }

// the lane's B-operand chunk: 8 input channels of its pixel, the producer's BatchNorm applied, rounded to bf16
__device__ __forceinline__ bf16x8 b_operand(const u32x4 raw, bool affine, const float* sc, const float* sh) {
  if (!affine) return __builtin_bit_cast(bf16x8, raw);
  float x[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __uint_as_float(raw[i] << 16);
    x[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], sc[e], sh[e]);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack2bf(x[2 * i], x[2 * i + 1]);
  return __builtin_bit_cast(bf16x8, o);
}

// ------------------------------------------------------------------------------------------------------------------
// statistics of the expansion output (sum, sum of squares of the bf16-rounded values per channel) without storing it
template <int NG>      // channel groups per workgroup: the block input is loaded and transformed once for all of them
__global__ __launch_bounds__(THREADS, 2) void k_exp_stats(const Args a) {
  __shared__ float red[4][2][NG * GC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, q = lane >> 4;
  const int p = blockIdx.x;
  const int cin = a.in.c;
  const bool kq = 8 * q < cin;
  const bool affine = a.in.scale != nullptr;
  bf16x8 afr[NG][3];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      afr[g][j] = kq ? *reinterpret_cast<const bf16x8*>(a.wt + (size_t)(g * GC + 16 * j + px) * a.ldw + 8 * q) : zero_frag();
  float isc[8], ish[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { isc[e] = 1.f; ish[e] = 0.f; }
  if (affine && kq) { loadf8(a.in.scale + 8 * q, isc); loadf8(a.in.shift + 8 * q, ish); }
  // Two 16-pixel chunks at a time: v_cvt_pk_bf16_f32 packs the SAME channel of the two chunks' pixels, and one
  // v_dot2c_f32_bf16 each adds the pair (against (1, 1)) and its squares to the channel's sums: 1.5 VALU instructions
  // per element instead of 2.5 with packed fp32 math (r06 lab: the pass is VALU-bound, 0.25 ms for 0.42 GB read)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  float s1[NG][3][4], s2[NG][3][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[g][j][r] = 0.f; s2[g][j][r] = 0.f; }
  const bf2_t ones = __builtin_bit_cast(bf2_t, 0x3f803f80u);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.in.data);
  const long long M = a.M, nchunks = (M + 15) / 16;
  const long long stride = (long long)a.P * 4;
  constexpr int UN = NG >= 3 ? 2 : 4;
  auto fetch = [&](long long c0, u32x4 (&raw)[UN], bool (&ok)[UN]) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long row = (c0 + u * stride) * 16 + px;
      ok[u] = row < M && kq;
      const long long rc = row < M ? row : M - 1;
      raw[u] = *reinterpret_cast<const u32x4*>(X + rc * a.in.ld + (kq ? 8 * q : 0));
    }
  };
  u32x4 raw[UN], nxt[UN];
  bool ok[UN], nok[UN];
  long long c0 = (long long)p * 4 + wave;
  if (c0 < nchunks) fetch(c0, raw, ok);
  for (; c0 < nchunks; c0 += stride * UN) {
    const long long c1 = c0 + stride * UN;
    fetch(c1 < nchunks ? c1 : c0, nxt, nok);          // the next round's loads under this round's arithmetic
#pragma unroll
    for (int u = 0; u < UN; u += 2) {
      const bf16x8 b0 = ok[u] ? b_operand(raw[u], affine, isc, ish) : zero_frag();
      const bf16x8 b1 = ok[u + 1] ? b_operand(raw[u + 1], affine, isc, ish) : zero_frag();
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[g][j], b0, z4, 0, 0, 0);
          const f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[g][j], b1, z4, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bf2_t pr = __builtin_bit_cast(bf2_t, pack2bf(e0[r], e1[r]));
            s1[g][j][r] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s1[g][j][r], false);
            s2[g][j][r] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2[g][j][r], false);
          }
        }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) { raw[u] = nxt[u]; ok[u] = nok[u]; }
  }
  // the 16 pixel lanes of a channel: xor butterfly (a fixed tree), then the waves in wave order
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[4] = {s1[g][j][2 * h], s1[g][j][2 * h + 1], s2[g][j][2 * h], s2[g][j][2 * h + 1]};
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], off, 64);
        if (px == 0) {
          const int c = g * GC + 16 * j + 4 * q + 2 * h;
          red[wave][0][c] = v[0]; red[wave][0][c + 1] = v[1];
          red[wave][1][c] = v[2]; red[wave][1][c + 1] = v[3];
        }
      }
  __syncthreads();
  for (int i = tid; i < 2 * NG * GC; i += THREADS) {
    const int r = i / (NG * GC), c = i % (NG * GC);
    const float t = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
    a.stat_partials[((size_t)p * 2 + r) * a.cexp + c] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Workgroup shape: NGR channel groups of 48 x NPH sixteen-pixel window parts = NGR * NPH waves.  A workgroup owns ALL
// expanded channels of its window, so the raw expanded row it stores (training) is contiguous in memory: 32 pixels x
// 192 / 288 bytes.  (r06 lab, 320x320x16->96: with the two channel groups of a pixel in two workgroups -- 96-byte
// pieces, two writers per 128-byte line -- the stores cost 0.44 ms of a 1.41 ms kernel; one group alone, contiguous
// rows: 0.07 ms for half the bytes; a plain fill writes the 2.5 GB in 0.37 ms.)
template <int NGR_, int NPH_> struct Shape {
  static constexpr int NGR = NGR_, NPH = NPH_;
  static constexpr int NW = NGR * NPH, NT = 64 * NW;
  static constexpr int WINC = 16 * NPH;       // window columns
  static constexpr int GCW = GC * NGR;        // channels of the workgroup (= cexp)
  static constexpr int GCP = GCW + 4;         // floats per pixel in the ring: the 8 lanes of a ds_write_b128 group on 32 banks
  static constexpr int ESTG = GCW * 2 + 8;    // bytes per pixel of the staged raw row: the 16 lanes of a ds_write_b64 group on 32 banks
  static constexpr int ECH = GCW * 2 / 16;    // 16-byte chunks per pixel
  static constexpr int TABF = 2 * GCW + 64;   // floats of the coefficient tables behind the ring
  static constexpr int NIT = (WINC * ECH + NT - 1) / NT;     // staged chunks per thread and row
};
template <int K, int S, typename SH> struct Geo {
  static constexpr int TXV = (SH::WINC - K) / S + 1;    // output columns whose taps lie inside the window
  static constexpr int NSL = (K + S - 1) / S;           // output rows in flight
  static constexpr int U0 = S * NSL;
  static constexpr int CPT = K == 3 ? 4 : 2;            // channels per tap thread (K*K*CPT/2 weight register pairs)
  static constexpr int NCH = SH::GCW / CPT;
  static constexpr int SLOTS = SH::NT / NCH;            // output pixels handled side by side
  static constexpr int NPX = (TXV + SLOTS - 1) / SLOTS;
  static constexpr int NF = U0;                         // rows of block-input loads in flight + 1 (16 bytes per lane and row)
};

// ACTM: 1 swish, 2 relu / relu6 / hswish / mish / srelu
// (Three waves per SIMD: at four, the 3x3 stride-2 instantiations fit 128 VGPRs only with 1-7 spilled registers, and the
// six-wave one then stored wrong values in the first row of a tile -- zeros in two of a lane's four MFMA results, r06
// scripts/mbf_debug.py; no instantiation may spill: tests/test_abi.py checks the build's resource report.)
template <int K, int S, int ACTM, bool STORE_E, typename SH>
__global__ __launch_bounds__(SH::NT, 3) void k_exp_dw_fwd(const Args a) {
  using G = Geo<K, S, SH>;
  constexpr int CPT = G::CPT, NV = CPT / 2, NSL = G::NSL, U0 = G::U0, NF = G::NF, PF = NF - 1;
  constexpr int U = U0 * NF / gcd_(U0, NF);
  constexpr int NPX = G::NPX, TXV = G::TXV, NCH = G::NCH, SLOTS = G::SLOTS;
  constexpr int NT = SH::NT, WINC = SH::WINC, GCW = SH::GCW, GCP = SH::GCP, ESTG = SH::ESTG, ECH = SH::ECH, NIT = SH::NIT;
  extern __shared__ float ring[];      // [2][WINC][GCP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, q = lane >> 4;
  const int gw = wave % SH::NGR, pw = wave / SH::NGR;      // the wave's channel group and window part
  const int H = a.in.h, W = a.in.w, cin = a.in.c;
  // block b runs on XCD b % 8: every XCD walks a contiguous range of tiles (neighbouring windows and row tiles share
  // block-input lines in that XCD's L2; dw_march.hip, r06)
  // (a.ngb > 1: the workgroup owns GCW of the cexp channels; the channel blocks of a tile slot sit back to back on one XCD)
  const int x8 = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int gblk = jb % a.ngb, ph = jb / a.ngb;
  const int cb = gblk * GCW;           // first channel of the workgroup
  const int t8 = (a.ntiles + 7) / 8;
  const int tile0 = x8 * t8 + ph, tstep = a.P / 8, tend = min(a.ntiles, (x8 + 1) * t8);
  const int pslot = ph * 8 + x8;
  const int cg0 = gw * GC;
  // ---- expansion-phase constants
  const bool kq = 8 * q < cin;
  const bool affine = a.in.scale != nullptr;
  bf16x8 afr[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    afr[j] = kq ? *reinterpret_cast<const bf16x8*>(a.wt + (size_t)(cb + cg0 + 16 * j + px) * a.ldw + 8 * q) : zero_frag();
  // coefficient tables in LDS (behind the ring): expansion BatchNorm of the workgroup's channels, the producer's
  // BatchNorm of the <= 32 block-input channels -- 40 VGPRs that buy another wave per SIMD
  float* tab = ring + 2 * WINC * GCP;      // [0,GCW) esc  [GCW,2 GCW) esh  then 32 isc, 32 ish
  // training: the raw bf16 row of the expanded tensor is staged here ([2][WINC][ESTG bytes]) and leaves as 16-byte
  // stores of whole pixels (r06 lab: 8-byte stores straight from the MFMA layout doubled the kernel's time)
  unsigned char* stage = reinterpret_cast<unsigned char*>(tab + SH::TABF);
  for (int i = tid; i < GCW; i += NT) { tab[i] = a.esc[cb + i]; tab[GCW + i] = a.esh[cb + i]; }
  if (tid < 32) {
    tab[2 * GCW + tid] = (affine && tid < cin) ? a.in.scale[tid] : 1.f;
    tab[2 * GCW + 32 + tid] = (affine && tid < cin) ? a.in.shift[tid] : 0.f;
  }
  __syncthreads();
  const float* t_esc = tab + cg0 + 4 * q;
  const float* t_isc = tab + 2 * GCW + (kq ? 8 * q : 0);
  const int wcol = pw * 16 + px;
  // store phase: thread -> up to NIT (pixel, 16-byte chunk) pairs of the staged row
  int spx[NIT], spart[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + NT * it;
    spx[it] = min(id / ECH, WINC - 1); spart[it] = id % ECH;      // (ids beyond the row: a valid LDS address, never stored)
  }
  // ---- tap-phase constants
  const int chunk = tid % NCH, slot = tid / NCH;
  const bool tact = tid < SLOTS * NCH;
  const int ct = chunk * CPT;
  f2 w[K * K][NV], st[2][NV];
  const f2 zero2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int i = 0; i < NV; ++i)
      w[t][i] = tact ? *reinterpret_cast<const f2*>(a.dww + (size_t)t * a.cexp + cb + ct + 2 * i) : zero2;
#pragma unroll
  for (int i = 0; i < NV; ++i) st[0][i] = st[1][i] = zero2;
  const bool want_stats = a.stat_partials != nullptr;
  const uint32_t irow_b = (uint32_t)W * a.in.ld * 2, orow_b = (uint32_t)a.ow * a.ldo * 2;
  const uint32_t erow_b = STORE_E ? (uint32_t)W * a.lde * 2 : 0u;
  const char* IN = reinterpret_cast<const char*>(a.in.data);
  char* OUT = reinterpret_cast<char*>(a.out);
  char* EO = reinterpret_cast<char*>(a.e_out);

  for (int tile = tile0; tile < tend; tile += tstep) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty = rr / a.tiles_x, tx = rr - ty * a.tiles_x;
    const int oy0 = ty * a.TY, oy1 = min(a.oh, oy0 + a.TY);
    const int wc0 = tx * TXV * S - a.pad_l;            // input column of window column 0
    const int ix = wc0 + wcol;
    const bool colok = ix >= 0 && ix < W;
    const uint32_t xoff = (uint32_t)min(max(ix, 0), W - 1) * (uint32_t)(a.in.ld * 2) + (kq ? 16u * q : 0u);
    // a window column is STORED by the tile that owns it: the first TXV * S columns, the whole window in the last tile
    uint32_t eoff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ixs = wc0 + spx[it];
      const bool eok = STORE_E && a.dbg != 1 && tid + NT * it < WINC * ECH && ixs >= 0 && ixs < W &&
                       (spx[it] < TXV * S || tx == a.tiles_x - 1);
      eoff[it] = eok ? (uint32_t)ixs * (uint32_t)(a.lde * 2) + (uint32_t)(cb * 2 + 16 * spart[it]) : OOB;
    }
    bool pok[NPX];
    uint32_t ooff[NPX], lrd[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int oxl = slot + i * SLOTS, ox = tx * TXV + oxl;
      pok[i] = tact && oxl < TXV && ox < a.ow;
      ooff[i] = pok[i] ? (uint32_t)ox * (uint32_t)(a.ldo * 2) + (uint32_t)(cb + ct) * 2u : OOB;
      lrd[i] = (uint32_t)(min(oxl, TXV - 1) * S * GCP + ct);
    }
    const brsrc_t in_img = make_rsrc(IN + (size_t)n * H * irow_b);
    const brsrc_t out_img = make_rsrc(OUT + (size_t)n * a.oh * orow_b);
    char* e_base = STORE_E ? EO + (size_t)n * H * erow_b : OUT;
    // steps t = input row + pad_t; the input row exists for pad_t <= t < H + pad_t; output row oy completes at oy S + K - 1
    const int t0 = oy0 * S, t_last = (oy1 - 1) * S + K - 1;
    URange row_rng, out_rng, erow_rng;
    row_rng.set(max(t0, a.pad_t), min(t_last + 1, H + a.pad_t));
    out_rng.set(t0 + K - 1, t_last + 1);
    erow_rng.set(max(t0, a.pad_t), min(ty == a.tiles_y - 1 ? t_last + 1 : oy1 * S, H + a.pad_t));
    f2 acc[NSL][NPX][NV];
#pragma unroll
    for (int s2 = 0; s2 < NSL; ++s2)
#pragma unroll
      for (int i = 0; i < NPX; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[s2][i][v] = zero2;
    u32x4 fx[NF];
    auto load_row = [&](int t) -> u32x4 {
      const uint32_t ro = (uint32_t)min(max(t - a.pad_t, 0), H - 1) * irow_b;
      return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_img, xoff, ro, 0));
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fx[i] = load_row(t0 + i);
    // U steps, statically unrolled.  The first round is peeled (called once in front of the loop): the compiler's count of
    // memory operations in flight at the loop head is the minimum over the entry edge and the back edge, and with the
    // prologue (PF loads, no stores) as the entry edge every round's first wait on the FIFO became vmcnt(1) -- a drain
    // of all stores and of the rows loaded ahead, once per round (r06, ISA of the first version)
    auto round = [&](const int tb) __attribute__((always_inline)) {
#pragma unroll
      for (int tt = 0; tt < U; ++tt) {
        const int t = tb + tt;
        fx[(tt + PF) % NF] = load_row(t + PF);
        const bool row_ok = row_rng.has(t);          // uniform
        float* buf = ring + (t & 1) * WINC * GCP;
        if (row_ok) {
          const bf16x8 b = kq ? b_operand(fx[tt % NF], affine, t_isc, t_isc + 32) : zero_frag();
          unsigned char* sbuf = stage + (t & 1) * WINC * ESTG + wcol * ESTG + cg0 * 2 + 8 * q;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j], b, z4, 0, 0, 0);
            u32x2 u;
            u[0] = pack2bf(e[0], e[1]);
            u[1] = pack2bf(e[2], e[3]);
            if (STORE_E) {
              *reinterpret_cast<u32x2*>(sbuf + 32 * j) = u;
            }
            float y[4];
            y[0] = __uint_as_float(u[0] << 16); y[1] = __uint_as_float(u[0] & 0xffff0000u);
            y[2] = __uint_as_float(u[1] << 16); y[3] = __uint_as_float(u[1] & 0xffff0000u);
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(t_esc + 16 * j);
            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(t_esc + GCW + 16 * j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float z = fmaf(y[r], sc4[r], sh4[r]);
              const float v = ACTM == 1 ? z * sigmoidf_(z) : act_other_(a.act, z);
              y[r] = colok ? v : 0.f;                  // 'SAME' padding is zero in the activated domain
            }
            *reinterpret_cast<float4*>(buf + wcol * GCP + cg0 + 16 * j + 4 * q) = make_float4(y[0], y[1], y[2], y[3]);
          }
        }
        __syncthreads();
        if (STORE_E) {
          const bool est = erow_rng.has(t);       // uniform; erow_rng lies inside row_rng
          char* e_row = e_base + (est ? (size_t)(t - a.pad_t) * erow_b : 0);
          const unsigned char* sb = stage + (t & 1) * WINC * ESTG;
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(sb + spx[it] * ESTG + 16 * spart[it]);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(sb + spx[it] * ESTG + 16 * spart[it] + 8);
            // A GLOBAL 16-byte store under the lane's own condition (exec mask, no branch: the block is one instruction),
            // not a buffer store.  buffer_store_dwordx4 -- which the backend also makes out of two adjacent 8-byte buffer
            // stores -- fetches its data registers late: the compiler put a VALU write of the store's FIRST data register
            // (v_cndmask of the next store's offset) two instructions behind the store, and on the device lanes 12-15 of
            // every 16-lane row then stored the new value as their first dword: wrong first channels of a chunk in the
            // expanded tensor, on some rows of the batch-128 run only, the depthwise output untouched (r06,
            // scripts/mbf_debug2.py; LLVM guards this store-data hazard only for MUBUF stores WITHOUT a register soffset).
            // Two separated 8-byte buffer stores are correct but cost 12-20 % of the kernel.
            u32x4 v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
            if (est && eoff[it] != OOB) *reinterpret_cast<u32x4*>(e_row + eoff[it]) = v;
          }
        }
        if (row_ok && tact) {
#pragma unroll
          for (int i = 0; i < NPX; ++i) {
            f2 x[K][NV];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              if (NV == 2) {
                const float4 t4 = *reinterpret_cast<const float4*>(buf + lrd[i] + kx * GCP);
                x[kx][0] = f2{t4.x, t4.y}; x[kx][NV - 1] = f2{t4.z, t4.w};
              } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) x[kx][v] = *reinterpret_cast<const f2*>(buf + lrd[i] + kx * GCP + 2 * v);
              }
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
              for (int ky = 0; ky < K; ++ky) {
                if (((tt - ky) % S + S) % S == 0) {          // static: this input row feeds output row (t - ky) / S
                  const int sl = slot_of(fdiv_(tt - ky, S), NSL);
#pragma unroll
                  for (int v = 0; v < NV; ++v) acc[sl][i][v] = w[ky * K + kx][v] * x[kx][v] + acc[sl][i][v];
                }
              }
            }
          }
        }
        if (((tt - (K - 1)) % S + S) % S == 0) {           // static: output row (t - K + 1) / S is complete
          const int sl = slot_of(fdiv_(tt - (K - 1), S), NSL);
          {
            const bool oyok = out_rng.has(t);      // uniform
            const uint32_t oro = oyok ? (uint32_t)((t - (K - 1)) / S) * orow_b : 0u;
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
              const bool valid = oyok && pok[i];
              uint32_t o[NV];
#pragma unroll
              for (int v = 0; v < NV; ++v) o[v] = pack2bf(acc[sl][i][v].x, acc[sl][i][v].y);
              const uint32_t oo = oyok ? ooff[i] : OOB;
              if (NV == 2) {
                u32x2 o2; o2[0] = o[0]; o2[1] = o[NV - 1];
                __builtin_amdgcn_raw_buffer_store_b64(o2, out_img, oo, oro, 0);
              } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) __builtin_amdgcn_raw_buffer_store_b32(o[v], out_img, oo + 4u * v, oro, 0);
              }
              if (want_stats) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                  const uint32_t ov = valid ? o[v] : 0u;
                  f2 r2;
                  r2.x = __uint_as_float(ov << 16); r2.y = __uint_as_float(ov & 0xffff0000u);
                  st[0][v] += r2;
                  st[1][v] = r2 * r2 + st[1][v];
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < NPX; ++i)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[sl][i][v] = zero2;
        }
      }
    };
    round(t0);
    for (int tb = t0 + U; tb <= t_last; tb += U) round(tb);
    __syncthreads();      // the next tile's first row reuses the ring slot of this tile's last rows
  }
  if (want_stats) {
    // per-thread channel sums -> one partial row per tile slot: thread-major in LDS, then the SLOTS pixel slots of a
    // channel in slot order (no atomics: the same sums on every run)
    float* red = ring;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      __syncthreads();
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        red[tid * CPT + 2 * v] = tact ? st[r][v].x : 0.f;
        red[tid * CPT + 2 * v + 1] = tact ? st[r][v].y : 0.f;
      }
      __syncthreads();
      for (int c = tid; c < GCW; c += NT) {
        const int ch = c / CPT, e = c % CPT;
        float t = 0.f;
        for (int s = 0; s < SLOTS; ++s) t += red[(s * NCH + ch) * CPT + e];
        a.stat_partials[((size_t)pslot * 2 + r) * a.cexp + cb + c] = t;
      }
    }
  }
}

// Tile slots (= workgroups = statistic partial rows, <= EDET_MAX_PARTS): at most ONE round of resident workgroups (r06
// lab, 160x160x24->144: 768 six-wave workgroups on a chip that holds 512 ran a second, half-empty round: +30 %), and a
// whole number of tiles per slot on every XCD.  EDET_MBF_P overrides the cap (lab switch).
inline void pick_slots(Args& a, const void* fn, int threads, size_t lds) {
  const char* p_env = getenv("EDET_MBF_P");
  int pmax = EDET_MAX_PARTS;
  if (p_env && p_env[0]) pmax = atoi(p_env);
  else {
    const int res = edet_resident_wgs(fn, threads, lds);
    if (res / a.ngb >= 8 && res / a.ngb < pmax) pmax = res / a.ngb;
  }
  if (pmax < 8) pmax = 8;
  const int t8 = (a.ntiles + 7) / 8;
  const int rounds = (t8 + pmax / 8 - 1) / (pmax / 8);
  a.P = 8 * ((t8 + rounds - 1) / rounds);
}

inline bool supported(const edet_tview_t* in, int cexp, int k, int s, int dtype) {
  if (dtype != EDET_BF16 || !in) return false;
  if (in->gate || in->act != EDET_ACT_NONE) return false;
  if (in->c % 8 != 0 || in->c > 32 || in->ld != in->c) return false;
  if (cexp % GC != 0 || cexp > 3 * GC) return false;      // a workgroup owns every expanded channel: up to three groups of 48
  // (k, stride) of the MBConv stages whose block input has <= 32 channels: 3x3 stride 1 / 2, 5x5 stride 2
  if (!((k == 3 && (s == 1 || s == 2)) || (k == 5 && s == 2))) return false;
  if ((int64_t)in->h * in->w * cexp * 2 >= 0x7fffffffLL) return false;      // 32-bit offsets inside one image
  return true;
}

}  // namespace mbf

/* 1 = the fused MBConv head kernels apply to this layer, 0 = use edet_pw_fwd + edet_dw_fwd */
extern "C" int edet_mbconv_fused_supported(const edet_tview_t* in, int cexp, int k, int stride, int dtype) {
  return mbf::supported(in, cexp, k, stride, dtype) ? 1 : 0;
}

extern "C" int edet_mbconv_expand_stats(const edet_tview_t* in, const void* wt, int ldw, int cexp,
                                        float* stat_partials, int* nparts_out, int dtype, void* stream) {
  using namespace mbf;
  EDET_CHECK(in && wt && stat_partials && nparts_out, "edet_mbconv_expand_stats: null argument");
  EDET_CHECK(supported(in, cexp, 3, 1, dtype), "edet_mbconv_expand_stats: unsupported layer (bf16, <= 32 input channels, "
             "expanded channels %% 48 == 0, plain affine input view)");
  EDET_CHECK(ldw >= in->c && ldw % 8 == 0, "edet_mbconv_expand_stats: bad kernel stride");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.wt = reinterpret_cast<const bf16_t*>(wt); a.ldw = ldw; a.cexp = cexp; a.stat_partials = stat_partials;
  a.ngroups = cexp / GC;
  a.M = (long long)in->n * in->h * in->w;
  const long long nchunks = (a.M + 15) / 16;
  long long P = (nchunks + 4 * 4 - 1) / (4 * 4);      // >= 4 chunks of 16 pixels per wave
  if (P > 1024) P = 1024;
  if (P < 1) P = 1;
  a.P = (int)P;
  const dim3 grid(a.P), block(THREADS);
  switch (a.ngroups) {
    case 1: edet_launch(k_exp_stats<1>, grid, block, 0, to_stream(stream), a); break;
    case 2: edet_launch(k_exp_stats<2>, grid, block, 0, to_stream(stream), a); break;
    default: edet_launch(k_exp_stats<3>, grid, block, 0, to_stream(stream), a); break;
  }
  EDET_LAUNCH_CHECK("edet_mbconv_expand_stats");
  *nparts_out = a.P;
  return 0;
}

extern "C" int edet_mbconv_expand_dw_fwd(const edet_tview_t* in, const void* wt, int ldw, int cexp,
                                         const float* exp_scale, const float* exp_shift, int act,
                                         void* expanded_out, int lde, const float* dw_weight, int k, int stride,
                                         void* out, int ldo, float* stat_partials, int* nparts_out, int dtype,
                                         void* stream) {
  using namespace mbf;
  EDET_CHECK(in && wt && exp_scale && exp_shift && dw_weight && out, "edet_mbconv_expand_dw_fwd: null argument");
  EDET_CHECK(supported(in, cexp, k, stride, dtype), "edet_mbconv_expand_dw_fwd: unsupported layer (bf16, <= 32 input "
             "channels, expanded channels %% 48 == 0, k in {3,5}, stride in {1,2}, plain affine input view)");
  EDET_CHECK(act >= EDET_ACT_SWISH && act <= EDET_ACT_LAST, "edet_mbconv_expand_dw_fwd: the expansion has an activation");
  EDET_CHECK(ldw >= in->c && ldw % 8 == 0 && ldo >= cexp && ldo % 2 == 0 && (!expanded_out || (lde >= cexp && lde % 4 == 0)),
             "edet_mbconv_expand_dw_fwd: bad strides");
  EDET_CHECK(!stat_partials || nparts_out, "edet_mbconv_expand_dw_fwd: nparts_out is needed with stat_partials");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = *in; a.wt = reinterpret_cast<const bf16_t*>(wt); a.ldw = ldw; a.cexp = cexp;
  a.esc = exp_scale; a.esh = exp_shift; a.act = act;
  a.e_out = reinterpret_cast<bf16_t*>(expanded_out); a.lde = lde;
  a.dww = dw_weight; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.oh = same_out(in->h, stride); a.ow = same_out(in->w, stride);
  a.pad_t = same_pad_before(in->h, k, stride); a.pad_l = same_pad_before(in->w, k, stride);
  EDET_CHECK((int64_t)a.oh * a.ow * ldo * 2 < 0x7fffffffLL && (int64_t)in->h * in->w * (lde > 0 ? lde : 1) * 2 < 0x7fffffffLL,
             "edet_mbconv_expand_dw_fwd: image too large for 32-bit offsets");
  a.ngroups = cexp / GC;
  const bool se = expanded_out != nullptr;
  // Workgroup shape.  Two channel groups (cexp = 96) in training: ONE four-wave workgroup owns both groups of a 32-column
  // window and stores whole pixels of the expanded row (r06 lab 320x320x16->96: 1.30 -> 1.06 ms against two workgroups
  // that each write 96 of a pixel's 192 bytes).  Otherwise four-wave workgroups of 48 channels over a 64-column window:
  // fewest recomputed halo columns, no barrier across more than four waves (160x160x24->144, six-wave whole-pixel shape
  // against three workgroups per window: training 0.78 / 0.56 vs 0.64 / 0.54 ms for k3s1 / k5s2, inference 0.60 / 0.50 vs
  // 0.46 / 0.31).  EDET_MBF_SHAPE=1|2 forces the per-group / whole-pixel shape (lab switch).
  const char* sh_env = getenv("EDET_MBF_SHAPE");
  const bool whole = (sh_env && sh_env[0]) ? sh_env[0] == '2' : (se && a.ngroups == 2);
  const int ngr = (whole && a.ngroups > 1) ? a.ngroups : 1;
  a.ngb = a.ngroups / ngr;
  const int winc = ngr == 1 ? 64 : 32;      // Shape<1,4>, Shape<2,2>, Shape<3,2>
  const int txv = (winc - k) / stride + 1;
  {
    const int cap = a.oh >= 160 ? 80 : 40;
    const int nt = (a.oh + cap - 1) / cap;
    a.TY = (a.oh + nt - 1) / nt;
  }
  a.tiles_x = (a.ow + txv - 1) / txv;
  a.tiles_y = (a.oh + a.TY - 1) / a.TY;
  a.ntiles = in->n * a.tiles_x * a.tiles_y;
  { const char* d = getenv("EDET_MBF_DBG"); a.dbg = (d && d[0]) ? atoi(d) : 0; }
  hipStream_t st = to_stream(stream);
  const bool sw = act == EDET_ACT_SWISH;
#define MBF_GO2(K_, S_, SH_)                                                                    \
  do {                                                                                          \
    const size_t lds = (size_t)(2 * SH_::WINC * SH_::GCP + SH_::TABF) * sizeof(float) +         \
                       (se ? (size_t)2 * SH_::WINC * SH_::ESTG : 0);                            \
    pick_slots(a, reinterpret_cast<const void*>(sw ? (se ? k_exp_dw_fwd<K_, S_, 1, true, SH_> : k_exp_dw_fwd<K_, S_, 1, false, SH_>) \
                                                      : (se ? k_exp_dw_fwd<K_, S_, 2, true, SH_> : k_exp_dw_fwd<K_, S_, 2, false, SH_>)), \
               SH_::NT, lds);                                                                   \
    const dim3 grid(a.P * a.ngb), block(SH_::NT);                                               \
    if (sw) { if (se) edet_launch(k_exp_dw_fwd<K_, S_, 1, true, SH_>, grid, block, lds, st, a); \
              else edet_launch(k_exp_dw_fwd<K_, S_, 1, false, SH_>, grid, block, lds, st, a); } \
    else { if (se) edet_launch(k_exp_dw_fwd<K_, S_, 2, true, SH_>, grid, block, lds, st, a);    \
           else edet_launch(k_exp_dw_fwd<K_, S_, 2, false, SH_>, grid, block, lds, st, a); }    \
  } while (0)
  typedef Shape<1, 4> Sh1;
  typedef Shape<2, 2> Sh2;
  typedef Shape<3, 2> Sh3;
#define MBF_GO(K_, S_)                                  \
  do {                                                  \
    if (ngr == 1) MBF_GO2(K_, S_, Sh1);                 \
    else if (ngr == 2) MBF_GO2(K_, S_, Sh2);            \
    else MBF_GO2(K_, S_, Sh3);                          \
  } while (0)
  if (k == 3 && stride == 1) MBF_GO(3, 1);
  else if (k == 3 && stride == 2) MBF_GO(3, 2);
  else MBF_GO(5, 2);
#undef MBF_GO
#undef MBF_GO2
  EDET_LAUNCH_CHECK("edet_mbconv_expand_dw_fwd");
  if (nparts_out) *nparts_out = a.P;
  return 0;
}
