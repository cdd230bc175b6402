// Tiled MFMA kernels for the wide pointwise (1x1) convolutions of the bf16 path on gfx950.
//
// The 1x1 convolutions of the late EfficientNet stages (40x40 and 20x20 feature maps, 80..1152 channels
// on one side and 480..1152 on the other; D7x: up to 3840) have 100-600 FLOP per byte of the streamed
// operand: they are the only layers of this network where the matrix cores, not the HBM stream, set the
// pace, and the wave-private streaming kernels of pw_stream.hip (one MFMA tile per wave, weights resident
// in LDS) do not fit them.  These kernels are classic workgroup-tiled GEMMs, with the producer's BatchNorm /
// swish / SE gate (forward) or the BatchNorm backward (gradients) applied between the global load and the
// LDS store of the streamed operand:
//
//   k_big_gemm<BWD=false>  out[m][j]  = sum_k view(in)[m][k] * Wt[j][k] + bias[j]      (+ BN stat partials)
//   k_big_gemm<BWD=true>   d in[m][k] = sum_n dy[m][n] * W[k][n], chained through act'(z), the optional
//                          accumulate, the BN-backward sums and the SE dgate sums in the epilogue
//   k_big_wgrad            dW[k][n]   = sum_m view(in)[m][k] * dy[m][n]  (split over m, partials -> workspace)
//
// Tile 128 x 128 per 256-thread workgroup (2 x 2 waves, 64 x 64 per wave = 2 x 2 v_mfma_f32_32x32x16_bf16),
// reduction step 64, two LDS stages of 2 x 18 KB (rows padded to 144 B: 9 sixteen-byte slots, so the 16
// rows a ds_read_b128 lane group touches fall on 16 different slots), one register set of raw loads in
// flight across the MFMA phase (issue t+2 / write t+1 / compute t), one barrier per step, two workgroups
// per CU.  Blocks are ordered so that the column tiles of one row tile run back to back on ONE XCD
// (block b runs on XCD b % 8): the streamed operand is read from HBM once and re-read from that XCD's L2.
//
// Reference call sites replaced: tf.keras.layers.Conv2D 1x1 in efficientdet/backbone/efficientnet_model.py
// :304-312 (expand), :345-353 (project), efficientdet/tf2/efficientdet_keras.py:286-290 (resample) and their
// gradients (TF Conv2DBackpropInput / Conv2DBackpropFilter under tf.GradientTape, train_lib.py:623-669).
#include <stdlib.h>

#include "common.h"

namespace pwb {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int THREADS = 256;
constexpr int BM = 128, BJ = 128, BK = 64;
constexpr int LDT = BK * 2 + 16;             // staged tile row stride in bytes (144)
constexpr int TILE_BYTES = BM * LDT;         // one operand tile (18432)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // two stages (73728)
constexpr int LDC_BF = BJ * 2 + 16;          // bf16 C tile row stride (272)
constexpr int LDC_F32 = BJ * 4 + 16;         // fp32 C tile row stride (528)
static_assert(BM * LDC_F32 <= SMEM_BYTES, "fp32 C tile must fit in the staging buffers");

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

// the first `nvalid` (1..7) bf16 elements of a 16-byte chunk, the others zeroed (a reduction length that is not a multiple
// of 8: the chunk that straddles it carries padding columns, which may hold anything)
__device__ __forceinline__ uint4 keep_first(uint4 v, int nvalid) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (2 * i >= nvalid) w[i] = 0u;
    else if (2 * i + 1 >= nvalid) w[i] &= 0xffffu;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

struct GemmArgs {
  edet_tview_t tv;    // fwd: streamed operand.  bwd: the conv's input view (epilogue chain target)
  edet_gview_t gv;    // bwd: streamed operand (dy)
  const bf16_t* Bm;   // [J][ldb], reduction index contiguous
  int ldb;
  int M, R, J;        // rows, reduction length, output columns
  int hw;             // pixels per image
  int ntm, ntj;       // row tiles, column tiles
  int tpw;            // consecutive row tiles per workgroup
  int ngrp;           // row-tile groups = stat partial rows
  const float* bias;  // fwd
  bf16_t* out;        // fwd
  int ldo;
  edet_bwd_epi_t epi; // bwd
  float* stat_partials;
  // CONV (implicit GEMM of a dense ck x ck convolution, stride cs, TF 'SAME'): rows = output pixels,
  // reduction index = (ky*ck + kx)*cin + c; tv is the conv INPUT view [n][ih][iw][cin]
  int ck, cs, cin, ih, iw, cow, cohw, pad_t, pad_l;
};

// 64 x 64 per wave: acc[nj][mi] = D[i = output column within the 32-tile][j = row within the 32-tile]
__device__ __forceinline__ void mma_stage(const unsigned char* As, const unsigned char* Bs, int wm, int wj, int lane,
                                          int ksub, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    if (kk < ksub) {
      const int koff = (kk * 16 + h * 8) * 2;
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bs + (wj * 64 + r) * LDT + koff);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bs + (wj * 64 + 32 + r) * LDT + koff);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + r) * LDT + koff);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + 32 + r) * LDT + koff);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
    }
  }
}

// OACT: the view's activation is relu / relu6 / hswish (utils.activation_fn; a.tv.act carries the code) -- a template
// parameter, so that the swish / linear instantiations keep their code and registers
// F32OUT (forward only, r04): the output is stored as fp32 -- the class / box predict layers of the INFERENCE forward, whose
// bf16 storage alone costs 3e-3 of the logit range (scripts/precision_sweep.py); a.out then points at floats, ldo in
// floats, no statistics.
template <bool BWD, bool GBN, bool CONV = false, bool OACT = false, bool F32OUT = false>
__global__ __launch_bounds__(THREADS, 2) void k_big_gemm(const GemmArgs a) {
  static_assert(!F32OUT || (!BWD && !CONV), "fp32 output: plain forward only");
  // CONV && BWD: data gradient of the dense convolution -- rows = INPUT pixels, the streamed operand dy is
  // gathered at (iy + pad - ky) / s when that is an integer inside the dy image, reduction index (ky*k+kx)*cout+co
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wj = wave >> 1;
  // block -> (row-tile group, column tile): all column tiles of a group back to back on one XCD
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int jt = q % a.ntj;
  const int grp = (q / a.ntj) * 8 + xcd;
  if (grp >= a.ngrp) return;
  const int j0 = jt * BJ;
  const bool want_stats = a.stat_partials != nullptr;
  const bool want_gate = BWD && a.epi.dgate != nullptr;       // SE-gated input: the gradient of the gated value is stored as it is
  const bool gate_sums = want_gate && !(a.epi.flags & EDET_EPI_GATE_SUMS_LATER);      // ... and its sums are formed here (atomics)
  const bool swish = !OACT && a.tv.act == EDET_ACT_SWISH, affine = a.tv.scale != nullptr;
  constexpr bool other = OACT;
  const bool gated = !BWD && a.tv.gate != nullptr;

  // staging geometry: thread -> chunk column lc (8 reduction elements), rows lr + 32*i
  const int lc = tid & 7, lr = tid >> 3;
  // epilogue geometry: thread -> 8 output columns ec*8.., rows er + 16*i
  const int ec = tid & 15, er = tid >> 4;
  const int ej = j0 + ec * 8;
  const bool ecol_ok = ej < a.J;

  const bf16_t* SRC = reinterpret_cast<const bf16_t*>(BWD ? a.gv.dz : a.tv.data);
  const bf16_t* SRCY = reinterpret_cast<const bf16_t*>(a.gv.y);
  const int lds_src = BWD ? a.gv.ld : a.tv.ld;
  const int nk = (a.R + BK - 1) / BK;

  float tot1 = 0.f, tot2 = 0.f;     // thread j < BJ: running column sums of this workgroup (stat partials)

  int64_t arow[4];
  int aimg[4];
  int ciy[CONV ? 4 : 1], cix[CONV ? 4 : 1];          // CONV: top-left input pixel of the row's window
  unsigned cvalid = 0;                               // CONV: taps of the chunk in flight that hit the image
  uint4 ra[4], ry[GBN ? 4 : 1], rb[4];
  // staging rows of row tile mt_
  auto setup = [&](int mt_) {
    const int m0 = mt_ * BM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + lr + 32 * i, a.M - 1);     // rows past M re-read row M-1 (never stored)
      if (CONV) {
        const int img = m / a.cohw, rem = m - img * a.cohw;
        const int oy = rem / a.cow, ox = rem - oy * a.cow;
        ciy[i] = BWD ? oy + a.pad_t : oy * a.cs - a.pad_t;     // BWD: (row, col) + pad, the tap is subtracted
        cix[i] = BWD ? ox + a.pad_l : ox * a.cs - a.pad_l;
        arow[i] = (int64_t)img * a.ih * a.iw;           // pixel index of the gathered image's first pixel
        aimg[i] = img;
      } else {
        arow[i] = (int64_t)m * lds_src;
        aimg[i] = gated ? m / a.hw : 0;
      }
    }
  };
  {
    auto issue = [&](int kt) {
      const int k = kt * BK + lc * 8;
      const bool kok = k < a.R;
      int cky = 0, ckx = 0, cc = 0;
      if (CONV) {
        const int tap = k / a.cin;
        cc = k - tap * a.cin;
        cky = tap / a.ck;
        ckx = tap - cky * a.ck;
        cvalid = 0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = make_uint4(0, 0, 0, 0);
        if (GBN) ry[i] = make_uint4(0, 0, 0, 0);
        rb[i] = make_uint4(0, 0, 0, 0);
        if (kok) {
          if (CONV) {
            int iy, ix;
            bool hit;
            if (BWD) {
              const int ty = ciy[i] - cky, tx = cix[i] - ckx;
              iy = ty / a.cs; ix = tx / a.cs;
              hit = ty >= 0 && tx >= 0 && iy * a.cs == ty && ix * a.cs == tx && iy < a.ih && ix < a.iw;
            } else {
              iy = ciy[i] + cky; ix = cix[i] + ckx;
              hit = iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw;
            }
            if (hit) {
              cvalid |= 1u << i;
              const int64_t off = (arow[i] + (int64_t)iy * a.iw + ix) * lds_src + cc;
              ra[i] = *reinterpret_cast<const uint4*>(SRC + off);
              if (GBN) ry[i] = *reinterpret_cast<const uint4*>(SRCY + off);
            }
          } else {
            ra[i] = *reinterpret_cast<const uint4*>(SRC + arow[i] + k);
            if (GBN) ry[i] = *reinterpret_cast<const uint4*>(SRCY + arow[i] + k);
          }
          const int j = j0 + lr + 32 * i;
          if (j < a.J) rb[i] = *reinterpret_cast<const uint4*>(a.Bm + (size_t)j * a.ldb + k);
        }
      }
    };
    auto commit = [&](int kt, unsigned char* stage) {
      const int k = kt * BK + lc * 8;
      const bool kok = k < a.R;
      unsigned char* As = stage;
      unsigned char* Bs = stage + TILE_BYTES;
      float c0[8], c1[8], c2[8];
      const int kc = CONV ? k % a.cin : k;     // channel of the streamed operand this chunk starts at
      if (kok) {
        if (!BWD) {
          if (affine) { loadf8(a.tv.scale + kc, c0); loadf8(a.tv.shift + kc, c1); }
        } else if (GBN) {
          loadf8(a.gv.a + kc, c0); loadf8(a.gv.b + kc, c1); loadf8(a.gv.cc + kc, c2);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 v = ra[i];
        if (kok) {
          if (!BWD) {
            if (affine || swish || other || gated) {
              float x[8];
              unpack8(ra[i], x);
              if (affine) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], c0[e], c1[e]);
              }
              if (other) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = act_other_(a.tv.act, x[e]);
              } else if (swish) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
              }
              if (gated) {
                float gt[8];
                loadf8(a.tv.gate + (size_t)aimg[i] * (CONV ? a.cin : a.R) + kc, gt);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] *= gt[e];
              }
              v = pack8(x);
              // 'SAME' padding is zero in the activated domain
              if (CONV && !((cvalid >> i) & 1u)) v = make_uint4(0, 0, 0, 0);
            }
          } else if (GBN) {
            float x[8], y[8];
            unpack8(ra[i], x);
            unpack8(ry[i], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(c0[e], x[e], fmaf(c1[e], y[e], c2[e]));
            v = pack8(x);
            if (CONV && !((cvalid >> i) & 1u)) v = make_uint4(0, 0, 0, 0);   // taps outside dy contribute nothing
          }
        }
        uint4 bv = rb[i];
        if (BWD && !GBN && !CONV && kok && a.R - k < 8) {     // ragged reduction length (the 810-column predict layers)
          v = keep_first(v, a.R - k);
          bv = keep_first(bv, a.R - k);
        }
        *reinterpret_cast<uint4*>(As + (lr + 32 * i) * LDT + lc * 16) = v;
        *reinterpret_cast<uint4*>(Bs + (lr + 32 * i) * LDT + lc * 16) = bv;
      }
    };

  // Row tiles of this workgroup.  The first reduction stage of tile mt+1 is requested at the START of tile mt's epilogue
  // (the accumulators are in LDS by then, their registers are free), so that it is in flight under the epilogue instead
  // of being waited for at the top of the next tile.
  const int mt_end = min(a.ntm, (grp + 1) * a.tpw);
  bool prefetched = false;
  for (int mt = grp * a.tpw; mt < mt_end; ++mt) {
    const int m0 = mt * BM;
    if (!prefetched) setup(mt);
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

    // BWD: the saved conv input of this tile's epilogue (act', SE gate sums, BatchNorm-backward sums).  Requested
    // during the LAST reduction step -- the staging registers of the streamed operand are free by then -- so that it
    // is in flight under the last MFMAs, the C-tile round trip through LDS and its barrier instead of after them.
    uint4 xr[BWD ? BM / 16 : 1];
    const bool need_x = BWD && (swish || other || want_gate || want_stats);

    __syncthreads();                     // the previous row tile's epilogue is done with the LDS
    if (!prefetched) issue(0);
    commit(0, smem);
    if (nk > 1) issue(1);
    __syncthreads();
    for (int kt = 0; kt + 1 < nk; ++kt) {
      unsigned char* cur = smem + (kt & 1) * STAGE_BYTES;
      unsigned char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
      commit(kt + 1, nxt);
      if (kt + 2 < nk) issue(kt + 2);
      mma_stage(cur, cur + TILE_BYTES, wm, wj, lane, BK / 16, acc);
      __syncthreads();
    }
    {  // last reduction step, peeled: nothing is staged any more
      if (BWD) {
        const bf16_t* Xp = reinterpret_cast<const bf16_t*>(a.tv.data);
#pragma unroll
        for (int i = 0; i < BM / 16; ++i) {
          const int m = m0 + er + 16 * i;
          xr[i] = make_uint4(0, 0, 0, 0);
          if (need_x && ecol_ok && m < a.M) xr[i] = *reinterpret_cast<const uint4*>(Xp + (size_t)m * a.tv.ld + ej);
        }
      }
      const int kt = nk - 1;
      unsigned char* cur = smem + (kt & 1) * STAGE_BYTES;
      const int krem = a.R - kt * BK;
      mma_stage(cur, cur + TILE_BYTES, wm, wj, lane, krem >= BK ? BK / 16 : (krem + 15) / 16, acc);
      __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue through LDS
    const int r = lane & 31, h = lane >> 5;
    if constexpr (F32OUT) {
      // C tile as fp32 [BM][LDC_F32], bias added; stored unrounded
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = wj * 64 + nj * 32 + 8 * g + 4 * h;
          float b4[4] = {0.f, 0.f, 0.f, 0.f};
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j0 + ch + e < a.J) b4[e] = a.bias[j0 + ch + e];
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            *reinterpret_cast<float4*>(smem + (wm * 64 + mi * 32 + r) * LDC_F32 + ch * 4) =
                make_float4(acc[nj][mi][4 * g + 0] + b4[0], acc[nj][mi][4 * g + 1] + b4[1], acc[nj][mi][4 * g + 2] + b4[2],
                            acc[nj][mi][4 * g + 3] + b4[3]);
        }
      __syncthreads();
      if (mt + 1 < mt_end) { setup(mt + 1); issue(0); prefetched = true; } else prefetched = false;
      float* OF = reinterpret_cast<float*>(a.out);
#pragma unroll
      for (int i = 0; i < BM / 16; ++i) {
        const int row = er + 16 * i;
        const int m = m0 + row;
        if (ecol_ok && m < a.M) {
          const float4 d0 = *reinterpret_cast<const float4*>(smem + row * LDC_F32 + ec * 32);
          const float4 d1 = *reinterpret_cast<const float4*>(smem + row * LDC_F32 + ec * 32 + 16);
          *reinterpret_cast<float4*>(OF + (size_t)m * a.ldo + ej) = d0;
          *reinterpret_cast<float4*>(OF + (size_t)m * a.ldo + ej + 4) = d1;
        }
      }
    } else if (!BWD) {
      // C tile as bf16 [BM][LDC_BF]: bias added, rounded once
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = wj * 64 + nj * 32 + 8 * g + 4 * h;
          float b4[4] = {0.f, 0.f, 0.f, 0.f};
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j0 + ch + e < a.J) b4[e] = a.bias[j0 + ch + e];
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            uint2 pk;
            pk.x = pack2bf(acc[nj][mi][4 * g + 0] + b4[0], acc[nj][mi][4 * g + 1] + b4[1]);
            pk.y = pack2bf(acc[nj][mi][4 * g + 2] + b4[2], acc[nj][mi][4 * g + 3] + b4[3]);
            *reinterpret_cast<uint2*>(smem + (wm * 64 + mi * 32 + r) * LDC_BF + ch * 2) = pk;
          }
        }
      }
      __syncthreads();
      if (!CONV && mt + 1 < mt_end) { setup(mt + 1); issue(0); prefetched = true; } else prefetched = false;   // next tile's first stage
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
      for (int i = 0; i < BM / 16; ++i) {
        const int row = er + 16 * i;
        const int m = m0 + row;
        if (ecol_ok && m < a.M) {
          const uint4 v = *reinterpret_cast<const uint4*>(smem + row * LDC_BF + ec * 16);
          *reinterpret_cast<uint4*>(a.out + (size_t)m * a.ldo + ej) = v;
          if (want_stats) {
            float x[8];
            unpack8(v, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += x[e]; s2[e] = fmaf(x[e], x[e], s2[e]); }
          }
        }
      }
      if (want_stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);            // [2][16][BJ]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[er * BJ + ec * 8 + e] = s1[e];
          red[(16 + er) * BJ + ec * 8 + e] = s2[e];
        }
        __syncthreads();
        if (tid < BJ) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { tot1 += red[i * BJ + tid]; tot2 += red[(16 + i) * BJ + tid]; }
        }
      }
    } else {
      // C tile as fp32 [BM][LDC_F32]
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = wj * 64 + nj * 32 + 8 * g + 4 * h;
            *reinterpret_cast<float4*>(smem + (wm * 64 + mi * 32 + r) * LDC_F32 + ch * 4) =
                make_float4(acc[nj][mi][4 * g + 0], acc[nj][mi][4 * g + 1], acc[nj][mi][4 * g + 2],
                            acc[nj][mi][4 * g + 3]);
          }
      __syncthreads();
      if (!CONV && mt + 1 < mt_end) { setup(mt + 1); issue(0); prefetched = true; } else prefetched = false;   // next tile's first stage
      const bf16_t* X = reinterpret_cast<const bf16_t*>(a.tv.data);
      bf16_t* GO = reinterpret_cast<bf16_t*>(a.epi.gout);
      float sc[8], sh[8], s1[8], s2[8], gp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; s1[e] = s2[e] = gp[e] = 0.f; }
      if (ecol_ok && affine) { loadf8(a.tv.scale + ej, sc); loadf8(a.tv.shift + ej, sh); }
      // dgate sums: a tile inside one image (the common case) is reduced through LDS to one atomic per
      // channel; a tile that straddles images flushes per thread whenever the image changes
      const int timg0 = m0 / a.hw, timg1 = (min(m0 + BM, a.M) - 1) / a.hw;
      const bool single_img = timg0 == timg1;
      int gp_img = -1;
      auto flush_gate = [&]() {
        if (gp_img >= 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(&a.epi.dgate[(size_t)gp_img * a.J + ej + e], gp[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) gp[e] = 0.f;
      };
      // the 16 threads of one row group walk the rows er, er+16, ... (xr: requested in the last reduction step)
#pragma unroll
      for (int i = 0; i < BM / 16; ++i) {
        const int row = er + 16 * i;
        const int m = m0 + row;
        if (ecol_ok && m < a.M) {
          const float4 d0 = *reinterpret_cast<const float4*>(smem + row * LDC_F32 + ec * 32);
          const float4 d1 = *reinterpret_cast<const float4*>(smem + row * LDC_F32 + ec * 32 + 16);
          const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          float x[8], g[8];
          unpack8(xr[i], x);
          const size_t off = (size_t)m * a.tv.ld + ej;
          if (want_gate) {
            if (gate_sums) {
              if (!single_img) {
                const int img = m / a.hw;
                if (img != gp_img) { flush_gate(); gp_img = img; }
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float z = fmaf(x[e], sc[e], sh[e]);
                gp[e] = fmaf(d[e], other ? act_other_(a.tv.act, z) : (swish ? swishf_(z) : z), gp[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = d[e];
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
          *reinterpret_cast<uint4*>(GO + off) = pack8(g);
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += g[e]; s2[e] = fmaf(g[e], x[e], s2[e]); }
          }
        }
      }
      if (gate_sums) {
        if (single_img) {
          __syncthreads();
          float* red = reinterpret_cast<float*>(smem);          // [16][BJ]
#pragma unroll
          for (int e = 0; e < 8; ++e) red[er * BJ + ec * 8 + e] = gp[e];
          __syncthreads();
          if (tid < BJ && j0 + tid < a.J) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += red[i * BJ + tid];
            atomicAdd(&a.epi.dgate[(size_t)timg0 * a.J + j0 + tid], t);
          }
        } else if (ecol_ok) {
          flush_gate();
        }
      }
      if (want_stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);            // [2][16][BJ]
        float mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = 0.f; rs[e] = 0.f; }
        if (ecol_ok) { loadf8(a.epi.mean + ej, mu); loadf8(a.epi.rstd + ej, rs); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[er * BJ + ec * 8 + e] = s1[e];
          red[(16 + er) * BJ + ec * 8 + e] = rs[e] * (s2[e] - mu[e] * s1[e]);    // sum g * (x - mean) * rstd
        }
        __syncthreads();
        if (tid < BJ) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { tot1 += red[i * BJ + tid]; tot2 += red[(16 + i) * BJ + tid]; }
        }
      }
    }
  }
  }   // (scope of the staging lambdas)
  if (want_stats && tid < BJ && j0 + tid < a.J) {
    float* dst = a.stat_partials + (size_t)grp * 2 * a.J;
    dst[j0 + tid] = tot1;
    dst[a.J + j0 + tid] = tot2;
  }
}

// ---------------------------------------------------------------------------- weight gradient
// dW[k][n] = sum_m X[m][k] * dY[m][n].  The contraction runs over the rows, so both operands are staged
// TRANSPOSED ([channel][64 rows], row index contiguous): a staging task is an 8-row x 8-channel block that
// a thread loads as 8 x 16 B (one per row), transforms, transposes on 16-bit units with v_perm_b32 and
// writes as 8 x 16 B (one per channel).  Waves 0-1 stage X (activated view), waves 2-3 stage dY (BatchNorm
// backward on load); lane -> (row group l % 8, channel chunk l / 8) so that the 8 lanes of a ds_write_b128
// group fill one contiguous 128-byte LDS row.
struct WgArgs {
  edet_tview_t tv;    // conv input view, K = tv.c
  edet_gview_t gv;    // dy, N = gv.c
  float* ws;          // [S][K][N] fp32 partial sums (or dW itself when S == 1 and accumulate)
  int M, K, N, hw;
  int ntk, ntn;       // 128-wide tiles over K and N
  int S;              // row splits
  int rows_per_split; // multiple of 64
  // CONV (weight gradient of a dense ck x ck convolution): rows = OUTPUT pixels (dy rows); "channel" kk of the x
  // operand = (tap, c): the activated input pixel shifted by the tap (zero outside the image); K = ck*ck*cin
  int ck, cs, cin, ih, iw, cow, cohw, pad_t, pad_l;
};

__device__ __forceinline__ uint32_t perm_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t perm_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// p[r] = 8 bf16 channels of row r  ->  LDS rows (cb + e), 8 consecutive row indices starting at rb
__device__ __forceinline__ void store_transposed(unsigned char* tile, int cb, int rb, const uint4 (&p)[8]) {
  const uint32_t w[8][4] = {{p[0].x, p[0].y, p[0].z, p[0].w}, {p[1].x, p[1].y, p[1].z, p[1].w},
                            {p[2].x, p[2].y, p[2].z, p[2].w}, {p[3].x, p[3].y, p[3].z, p[3].w},
                            {p[4].x, p[4].y, p[4].z, p[4].w}, {p[5].x, p[5].y, p[5].z, p[5].w},
                            {p[6].x, p[6].y, p[6].z, p[6].w}, {p[7].x, p[7].y, p[7].z, p[7].w}};
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) {       // channel pair (2*c2, 2*c2+1) lives in word c2 of every row
    uint4 lo, hi;
    lo.x = perm_lo(w[0][c2], w[1][c2]); lo.y = perm_lo(w[2][c2], w[3][c2]);
    lo.z = perm_lo(w[4][c2], w[5][c2]); lo.w = perm_lo(w[6][c2], w[7][c2]);
    hi.x = perm_hi(w[0][c2], w[1][c2]); hi.y = perm_hi(w[2][c2], w[3][c2]);
    hi.z = perm_hi(w[4][c2], w[5][c2]); hi.w = perm_hi(w[6][c2], w[7][c2]);
    *reinterpret_cast<uint4*>(tile + (cb + 2 * c2) * LDT + rb * 2) = lo;
    *reinterpret_cast<uint4*>(tile + (cb + 2 * c2 + 1) * LDT + rb * 2) = hi;
  }
}

template <bool GBN, bool CONV = false>
__global__ __launch_bounds__(THREADS, 2) void k_big_wgrad(const WgArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wj = wave >> 1;    // MFMA: wm -> K half (first operand), wj -> N half
  const int ntile = a.ntk * a.ntn;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile = q % ntile;
  const int split = (q / ntile) * 8 + xcd;
  if (split >= a.S) return;
  const int kt0 = (tile / a.ntn) * 128, nt0 = (tile % a.ntn) * 128;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);

  // staging task of this thread
  const bool is_x = wave < 2;
  const int rg = lane & 7;                          // row group: rows rg*8 .. rg*8+7 of the 64-row step
  const int cchunk = (lane >> 3) + 8 * (wave & 1);  // channel chunk 0..15 of the 128-wide tile
  const int cb = cchunk * 8;
  const int cglob = (is_x ? kt0 : nt0) + cb;
  const bool c_ok = cglob < (is_x ? a.K : a.N);
  // CONV: tap and input channel of this thread's x chunk (cin % 8 == 0: a chunk never straddles taps)
  const int ctap = (CONV && is_x) ? cglob / a.cin : 0;
  const int cch = (CONV && is_x) ? cglob - ctap * a.cin : cglob;
  const int cky = CONV ? ctap / a.ck : 0, ckx = CONV ? ctap - (ctap / a.ck) * a.ck : 0;
  unsigned cvalid = 0;
  const bf16_t* SRC = reinterpret_cast<const bf16_t*>(is_x ? a.tv.data : a.gv.dz);
  const bf16_t* SRCY = reinterpret_cast<const bf16_t*>(a.gv.y);
  const int ld = is_x ? a.tv.ld : a.gv.ld;
  const bool swish = a.tv.act == EDET_ACT_SWISH, affine = a.tv.scale != nullptr, gated = a.tv.gate != nullptr;
  constexpr bool other = false;     // (the dense-convolution instantiations: swish / linear views only)
  float c0[8], c1[8], c2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { c0[e] = 1.f; c1[e] = 0.f; c2[e] = 0.f; }
  if (c_ok) {
    if (is_x) {
      if (affine) { loadf8(a.tv.scale + cch, c0); loadf8(a.tv.shift + cch, c1); }
    } else if (GBN) {
      loadf8(a.gv.a + cglob, c0); loadf8(a.gv.b + cglob, c1); loadf8(a.gv.cc + cglob, c2);
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

  uint4 ra[8], ry[GBN ? 8 : 1];
  auto issue = [&](int mb) {
    int cimg = 0, coy = 0, cox = 0;            // CONV: output pixel of row mb + rg*8, advanced row by row
    if (CONV) {
      cvalid = 0;
      const int m8 = mb + rg * 8;
      cimg = m8 / a.cohw;
      const int rem = m8 - cimg * a.cohw;
      coy = rem / a.cow;
      cox = rem - coy * a.cow;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mb + rg * 8 + r;
      ra[r] = make_uint4(0, 0, 0, 0);
      if (GBN) ry[r] = make_uint4(0, 0, 0, 0);
      if (CONV && r > 0) {
        if (++cox == a.cow) {
          cox = 0;
          if (++coy * a.cow == a.cohw) { coy = 0; ++cimg; }
        }
      }
      if (c_ok && m < m_end) {
        if (CONV && is_x) {
          const int iy = coy * a.cs - a.pad_t + cky, ix = cox * a.cs - a.pad_l + ckx;
          if (iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw) {
            cvalid |= 1u << r;
            ra[r] = *reinterpret_cast<const uint4*>(SRC + (((size_t)cimg * a.ih + iy) * a.iw + ix) * ld + cch);
          }
        } else {
          ra[r] = *reinterpret_cast<const uint4*>(SRC + (size_t)m * ld + cglob);
          if (GBN && !is_x) ry[r] = *reinterpret_cast<const uint4*>(SRCY + (size_t)m * ld + cglob);
        }
      }
    }
  };
  auto commit = [&](int mb, unsigned char* stage) {
    unsigned char* tile_lds = stage + (is_x ? 0 : TILE_BYTES);
    uint4 p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mb + rg * 8 + r;
      p[r] = ra[r];
      if (c_ok && m < m_end) {
        if (is_x) {
          if (affine || swish || other || gated) {
            float x[8];
            unpack8(ra[r], x);
            if (affine) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], c0[e], c1[e]);
            }
            if (other) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = act_other_(a.tv.act, x[e]);
            } else if (swish) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
            }
            if (gated) {
              float gt[8];
              loadf8(a.tv.gate + (size_t)(m / (CONV ? a.cohw : a.hw)) * (CONV ? a.cin : a.K) + cch, gt);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] *= gt[e];
            }
            p[r] = pack8(x);
          }
          if (CONV && !((cvalid >> r) & 1u)) p[r] = make_uint4(0, 0, 0, 0);   // 'SAME' padding: zero after act
        } else if (GBN) {
          float x[8], y[8];
          unpack8(ra[r], x);
          unpack8(ry[r], y);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaf(c0[e], x[e], fmaf(c1[e], y[e], c2[e]));
          p[r] = pack8(x);
        }
      } else {
        p[r] = make_uint4(0, 0, 0, 0);      // rows past the split / channels past K, N contribute zero
      }
    }
    store_transposed(tile_lds, cb, rg * 8, p);
  };

  const int nst = (m_end - m_begin + BK - 1) / BK;
  if (nst > 0) {
    issue(m_begin);
    commit(m_begin, smem);
    if (nst > 1) issue(m_begin + BK);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      unsigned char* cur = smem + (st & 1) * STAGE_BYTES;
      unsigned char* nxt = smem + ((st + 1) & 1) * STAGE_BYTES;
      if (st + 1 < nst) commit(m_begin + (st + 1) * BK, nxt);
      if (st + 2 < nst) issue(m_begin + (st + 2) * BK);
      mma_stage(cur + TILE_BYTES, cur, wj, wm, lane, BK / 16, acc);
      __syncthreads();
    }
  }
  // mma_stage(As := dY^T tile, Bs := X^T tile, wm := wj, wj := wm): its first MFMA operand comes from "Bs"
  // rows (wj_arg*64 + ...) = X^T rows = k, its second from "As" rows (wm_arg*64 + ...) = dY^T rows = n, so
  // acc[kk][nn] = D[i = k within the 32-tile][j = n within the 32-tile]: lane holds n = lane & 31 and
  // k = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
  float* dst = a.ws + (size_t)split * a.K * a.N;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int n = nt0 + wj * 64 + nn * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = kt0 + wm * 64 + kk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (k < a.K && n < a.N) dst[(size_t)k * a.N + n] = acc[kk][nn][e];
      }
    }
}

// ---- balanced staging variant (the pointwise weight gradient's default since r02a) ---------------------------------
// In k_big_wgrad waves 0-1 stage X (BatchNorm + swish + gate: ~10 issue slots per element) while waves 2-3 stage dY
// (BatchNorm backward: ~3), and all four meet at the barrier: the r01g SQ counters show the step time tracking the
// X waves.  Here every thread stages one 4-row x 8-channel block of X AND one of dY (256 blocks each per 64-row
// step), so the four waves carry equal work; LDS writes become 8-byte (4 rows) instead of 16-byte.
// p[r] = 8 bf16 channels of row r (4 rows)  ->  LDS rows (cb + e), 4 consecutive row indices starting at rb
__device__ __forceinline__ void store_transposed4(unsigned char* tile, int cb, int rb, const uint4 (&p)[4]) {
  const uint32_t w[4][4] = {{p[0].x, p[0].y, p[0].z, p[0].w}, {p[1].x, p[1].y, p[1].z, p[1].w},
                            {p[2].x, p[2].y, p[2].z, p[2].w}, {p[3].x, p[3].y, p[3].z, p[3].w}};
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) {
    uint2 lo, hi;
    lo.x = perm_lo(w[0][c2], w[1][c2]); lo.y = perm_lo(w[2][c2], w[3][c2]);
    hi.x = perm_hi(w[0][c2], w[1][c2]); hi.y = perm_hi(w[2][c2], w[3][c2]);
    *reinterpret_cast<uint2*>(tile + (cb + 2 * c2) * LDT + rb * 2) = lo;
    *reinterpret_cast<uint2*>(tile + (cb + 2 * c2 + 1) * LDT + rb * 2) = hi;
  }
}

template <bool GBN, bool OACT = false>
__global__ __launch_bounds__(THREADS, 2) void k_big_wgrad_bal(const WgArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wj = wave >> 1;
  const int ntile = a.ntk * a.ntn;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile = q % ntile;
  const int split = (q / ntile) * 8 + xcd;
  if (split >= a.S) return;
  const int kt0 = (tile / a.ntn) * 128, nt0 = (tile % a.ntn) * 128;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);

  // staging tasks of this thread: rows rg*4 .. rg*4+3 of the 64-row step, channel chunk cchunk of BOTH operands
  const int rg = lane & 15;
  const int cchunk = (lane >> 4) + 4 * wave;        // 0..15
  const int cb = cchunk * 8;
  const int xglob = kt0 + cb, dglob = nt0 + cb;
  const bool x_ok = xglob < a.K, d_ok = dglob < a.N;
  const bf16_t* XS = reinterpret_cast<const bf16_t*>(a.tv.data);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gv.dz);
  const bf16_t* DY = reinterpret_cast<const bf16_t*>(a.gv.y);
  const bool swish = !OACT && a.tv.act == EDET_ACT_SWISH, affine = a.tv.scale != nullptr, gated = a.tv.gate != nullptr;
  constexpr bool other = OACT;
  float xs[8], xt[8], ga[8], gb[8], gc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { xs[e] = 1.f; xt[e] = 0.f; ga[e] = 1.f; gb[e] = 0.f; gc[e] = 0.f; }
  if (x_ok && affine) { loadf8(a.tv.scale + xglob, xs); loadf8(a.tv.shift + xglob, xt); }
  if (d_ok && GBN) { loadf8(a.gv.a + dglob, ga); loadf8(a.gv.b + dglob, gb); loadf8(a.gv.cc + dglob, gc); }

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

  uint4 rx[4], rz[4], ry[GBN ? 4 : 1];
  auto issue = [&](int mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mb + rg * 4 + r;
      rx[r] = make_uint4(0, 0, 0, 0);
      rz[r] = make_uint4(0, 0, 0, 0);
      if (GBN) ry[r] = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (x_ok) rx[r] = *reinterpret_cast<const uint4*>(XS + (size_t)m * a.tv.ld + xglob);
        if (d_ok) {
          rz[r] = *reinterpret_cast<const uint4*>(DZ + (size_t)m * a.gv.ld + dglob);
          if (GBN) ry[r] = *reinterpret_cast<const uint4*>(DY + (size_t)m * a.gv.ld + dglob);
        }
      }
    }
  };
  auto commit = [&](int mb, unsigned char* stage) {
    uint4 px[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mb + rg * 4 + r;
      px[r] = make_uint4(0, 0, 0, 0);       // rows past the split / channels past K, N contribute zero
      pd[r] = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (x_ok) {
          px[r] = rx[r];
          if (affine || swish || other || gated) {
            float x[8];
            unpack8(rx[r], x);
            if (affine) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], xs[e], xt[e]);
            }
            if (other) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = act_other_(a.tv.act, x[e]);
            } else if (swish) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
            }
            if (gated) {
              float gt[8];
              loadf8(a.tv.gate + (size_t)(m / a.hw) * a.K + xglob, gt);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] *= gt[e];
            }
            px[r] = pack8(x);
          }
        }
        if (d_ok) {
          pd[r] = rz[r];
          if (GBN) {
            float x[8], y[8];
            unpack8(rz[r], x);
            unpack8(ry[r], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(ga[e], x[e], fmaf(gb[e], y[e], gc[e]));
            pd[r] = pack8(x);
          }
        }
      }
    }
    store_transposed4(stage, cb, rg * 4, px);                  // X^T tile
    store_transposed4(stage + TILE_BYTES, cb, rg * 4, pd);     // dY^T tile
  };

  const int nst = (m_end - m_begin + BK - 1) / BK;
  if (nst > 0) {
    issue(m_begin);
    commit(m_begin, smem);
    if (nst > 1) issue(m_begin + BK);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      unsigned char* cur = smem + (st & 1) * STAGE_BYTES;
      unsigned char* nxt = smem + ((st + 1) & 1) * STAGE_BYTES;
      if (st + 1 < nst) commit(m_begin + (st + 1) * BK, nxt);
      if (st + 2 < nst) issue(m_begin + (st + 2) * BK);
      mma_stage(cur + TILE_BYTES, cur, wj, wm, lane, BK / 16, acc);
      __syncthreads();
    }
  }
  float* dst = a.ws + (size_t)split * a.K * a.N;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int n = nt0 + wj * 64 + nn * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = kt0 + wm * 64 + kk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (k < a.K && n < a.N) dst[(size_t)k * a.N + n] = acc[kk][nn][e];
      }
    }
}

inline bool big_lds_ok(const void* kern) {
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES) == hipSuccess;
}

}  // namespace pwb

// consecutive row tiles per workgroup: as few as the partial-row limit allows; EDET_BIG_TPW = a minimum (lab switch)
static int big_tpw(int ntm) {
  int t = (ntm + EDET_MAX_PARTS - 1) / EDET_MAX_PARTS;
  const char* e = getenv("EDET_BIG_TPW");
  if (e && e[0] && atoi(e) > t) t = atoi(e);
  return t;
}

// the LDS-DMA forward kernel (pw_glds.hip); same return convention
int pwg_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, int tpw, hipStream_t st);
// Which wide forward layers go to the LDS-DMA kernel: the SE-gated views (the MBConv project layers: 5-19 % faster there,
// r06al); it ties or loses a few per cent on the others.  EDET_PW_GLDS = 0: never; 2: every shape of its envelope (read
// per call: lab switch and the bit-equality test).  The two kernels give the same bits, so the rule is free to change.
static bool glds_wanted(const edet_tview_t* in) {
  const char* e = getenv("EDET_PW_GLDS");
  const int mode = e && e[0] ? atoi(e) : 1;
  return mode == 2 || (mode == 1 && in->gate != nullptr);
}

// return 1 = handled, 0 = shape outside the envelope (caller falls back), < 0 = error
int pwb_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, hipStream_t st) {
  using namespace pwb;
  const int K = in->c, N = cout;
  if (K % 8 != 0 || in->ld % 8 != 0 || ldw % 8 != 0 || ldo % 8 != 0 || ldo < (N + 7) / 8 * 8) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = reinterpret_cast<const bf16_t*>(wt); a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = K; a.J = N; a.hw = in->h * in->w;
  a.bias = bias; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (N + BJ - 1) / BJ;
  a.tpw = big_tpw(a.ntm);
  if (glds_wanted(in)) {
    const int rc = pwg_try_fwd(in, wt, ldw, bias, out, cout, ldo, stat_partials, nparts_out, a.tpw, st);
    if (rc != 0) return rc;
  }
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  if (nparts_out) *nparts_out = a.ngrp;
  static const bool ok = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<false, false>)) &&
                         big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<false, false, false, true>));
  if (!ok) return 0;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  if (in->act > EDET_ACT_SWISH) edet_launch(k_big_gemm<false, false, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_gemm<false, false>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_pw_fwd(big)");
  return 1;
}

// forward with fp32 output [rows][ldo] (ldo in floats, >= cout rounded up to 8); no statistics
int pwb_fwd_f32out(const edet_tview_t* in, const void* wt, int ldw, const float* bias, float* out, int cout, int ldo,
                   hipStream_t st) {
  using namespace pwb;
  const int K = in->c, N = cout;
  if (K % 8 != 0 || in->ld % 8 != 0 || ldw % 8 != 0 || ldo % 4 != 0 || ldo < (N + 7) / 8 * 8) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = reinterpret_cast<const bf16_t*>(wt); a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = K; a.J = N; a.hw = in->h * in->w;
  a.bias = bias; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (N + BJ - 1) / BJ;
  a.tpw = big_tpw(a.ntm);
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  static const bool ok = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<false, false, false, false, true>)) &&
                         big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<false, false, false, true, true>));
  if (!ok) return 0;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  if (in->act > EDET_ACT_SWISH) edet_launch(k_big_gemm<false, false, false, true, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_gemm<false, false, false, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_pw_fwd_f32out");
  return 1;
}

// dense k x k convolution (stride s, TF 'SAME') as an implicit GEMM: wt [cout][k*k*cin], reduction index
// (ky*k + kx)*cin + c contiguous.  return 1 = handled, 0 = shape outside the envelope, < 0 = error
int pwb_try_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int s, const float* bias, void* out,
                     int cout, int ldo, float* stat_partials, int* nparts_out, hipStream_t st) {
  if (in->act > EDET_ACT_SWISH) return 0;     // relu / relu6 / hswish: the direct dense-convolution kernels (conv.hip)
  using namespace pwb;
  const int cin = in->c, N = cout;
  if (cin % 8 != 0 || in->ld % 8 != 0 || ldw % 8 != 0 || ldo % 8 != 0 || ldo < (N + 7) / 8 * 8) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = reinterpret_cast<const bf16_t*>(wt); a.ldb = ldw;
  const int oh = same_out(in->h, s), ow = same_out(in->w, s);
  a.M = in->n * oh * ow; a.R = k * k * cin; a.J = N; a.hw = oh * ow;
  a.ck = k; a.cs = s; a.cin = cin; a.ih = in->h; a.iw = in->w; a.cow = ow; a.cohw = oh * ow;
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  a.bias = bias; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (N + BJ - 1) / BJ;
  a.tpw = big_tpw(a.ntm);
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  if (nparts_out) *nparts_out = a.ngrp;
  static const bool ok = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<false, false, true>));
  if (!ok) return 0;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  edet_launch(k_big_gemm<false, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_conv_fwd(big)");
  return 1;
}

int pwb_try_dgrad(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st) {
  using namespace pwb;
  const int R = dy->c, KO = in->c;
  if (KO % 8 != 0 || in->ld % 8 != 0 || dy->ld % 8 != 0 || ldw % 8 != 0) return 0;
  // a reduction length that is not a multiple of 8 (the 810 / 36 columns of the predict layers) only without a
  // BatchNorm backward on dy (per-channel vectors are read in chunks of 8) and for the wide inputs the streaming
  // kernels cannot take (efficientdet-d3 and up: 160 .. 384 filters); the straddling chunk is masked
  if (R % 8 != 0 && (dy->a || KO < 160 || dy->ld < (R + 7) / 8 * 8 || ldw < (R + 7) / 8 * 8)) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy;
  a.Bm = reinterpret_cast<const bf16_t*>(w); a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = R; a.J = KO; a.hw = in->h * in->w;
  a.epi = *epi; a.stat_partials = epi->stat_partials;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (KO + BJ - 1) / BJ;
  a.tpw = big_tpw(a.ntm);
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  if (nparts_out) *nparts_out = a.ngrp;
  static const bool ok1 = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, false>)) &&
                          big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, false, false, true>));
  static const bool ok2 = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, true>)) &&
                          big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, true, false, true>));
  if (!ok1 || !ok2) return 0;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  if (in->act > EDET_ACT_SWISH) {
    if (dy->a) edet_launch(k_big_gemm<true, true, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
    else edet_launch(k_big_gemm<true, false, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  } else if (dy->a) edet_launch(k_big_gemm<true, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_gemm<true, false>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_pw_bwd_data(big)");
  return 1;
}

int pwb_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st) {
  using namespace pwb;
  const int K = in->c, N = dy->c;
  if (!workspace || K % 8 != 0 || dy->ld % 8 != 0 || in->ld % 8 != 0) return 0;
  // N % 8 != 0 (predict layers): every output column depends on its own dy column only and the stores are guarded per
  // element, so the padding columns of the straddling chunk never reach dW; as for the data gradient, only without a
  // BatchNorm backward on dy and for the wide inputs
  if (N % 8 != 0 && (dy->a || K < 160 || dy->ld < (N + 7) / 8 * 8)) return 0;
  WgArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy; a.ws = reinterpret_cast<float*>(workspace);
  a.M = in->n * in->h * in->w; a.K = K; a.N = N; a.hw = in->h * in->w;
  a.ntk = (K + 127) / 128; a.ntn = (N + 127) / 128;
  const int ntile = a.ntk * a.ntn;
  const int64_t kn = (int64_t)K * N;
  // Workgroup target, at least 4 steps of 64 rows each, bounded by the workspace.  Every split writes a K x N fp32
  // partial that edet_reduce_partials reads back, so more splits are not free: r03c lab (scripts/kernel_lab.py
  // --entry pw_bwd_weight --layers mid --ab EDET_WGRAD_WGS=2048,1024,512,256, D0 640x640 batch 128): the 16 mid-size
  // layers take 2.96 ms at 2048 workgroups (round 2), 2.59 ms with 512 for K*N < 64 K and 1024 above -- at 2048 the
  // partials of 1152 x 320 (76 splits, 112 MB written and read back) outweigh the 183 MB the kernel streams.
  // EDET_WGRAD_WGS overrides (lab switch, read per call).
  const char* wgs_env = getenv("EDET_WGRAD_WGS");
  const int wg_target = wgs_env ? atoi(wgs_env) : (kn >= 65536 ? 1024 : 512);
  // rounded DOWN: two workgroups of this kernel are resident per compute unit (512 at a time), and a grid of 513 or
  // 1026 (288 x 48: 3 tiles x 171 splits; 1152 x 320: 27 x 38) runs a last round for one or two stragglers
  // (r03e: efficientdet-d7x 384x384x288->48 went from 0.93 to 1.19 ms per call when the target dropped to 512)
  int S = wg_target / ntile;
  if (S < 1) S = 1;
  const int max_by_rows = (a.M + 4 * BK - 1) / (4 * BK);
  if (S > max_by_rows) S = max_by_rows;
  const int64_t max_by_ws = (int64_t)(workspace_bytes / sizeof(float)) / kn;
  if (S > max_by_ws) S = (int)max_by_ws;
  if (S < 1) return 0;
  a.rows_per_split = ((a.M + S - 1) / S + BK - 1) / BK * BK;
  a.S = (a.M + a.rows_per_split - 1) / a.rows_per_split;
  // the balanced-staging kernel (r02a, 17 mid-size layers of D0 at batch 128: 6.03 ms against 6.53 ms for the
  // two-waves-per-operand staging of k_big_wgrad, which remains for the dense-convolution variant)
  static const bool ok1 = big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad_bal<false>)) &&
                          big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad_bal<false, true>));
  static const bool ok2 = big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad_bal<true>)) &&
                          big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad_bal<true, true>));
  if (!ok1 || !ok2) return 0;
  const int grid = (a.S + 7) / 8 * 8 * ntile;
  if (in->act > EDET_ACT_SWISH) {
    if (dy->a) edet_launch(k_big_wgrad_bal<true, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
    else edet_launch(k_big_wgrad_bal<false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  } else if (dy->a) edet_launch(k_big_wgrad_bal<true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_wgrad_bal<false>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_pw_bwd_weight(big)");
  if (edet_reduce_partials(a.ws, a.S, kn, dweight, st) != 0) return -2;
  return 1;
}

// ---- dense k x k convolution backward as implicit GEMMs (conv.hip dispatches here for bf16) -------------------
// data gradient: w_t [cin][ldw] with the reduction index (ky*k + kx)*cout + co contiguous; rows = input pixels
int pwb_try_conv_dgrad(const edet_gview_t* dy, const void* w_t, int ldw, int k, int s, const edet_tview_t* in,
                       const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st) {
  if (in->act > EDET_ACT_SWISH) return 0;     // relu / relu6 / hswish: the direct dense-convolution kernels (conv.hip)
  using namespace pwb;
  const int cout = dy->c, cin = in->c;
  if (cout % 8 != 0 || cin % 8 != 0 || in->ld % 8 != 0 || dy->ld % 8 != 0 || ldw % 8 != 0) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy;
  a.Bm = reinterpret_cast<const bf16_t*>(w_t); a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = k * k * cout; a.J = cin; a.hw = in->h * in->w;
  a.ck = k; a.cs = s; a.cin = cout;                 // channels of the STREAMED operand (dy)
  a.ih = dy->h; a.iw = dy->w;                       // gathered image = dy
  a.cow = in->w; a.cohw = in->h * in->w;            // row space = input pixels
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  a.epi = *epi; a.stat_partials = epi->stat_partials;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (cin + BJ - 1) / BJ;
  a.tpw = big_tpw(a.ntm);
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  if (nparts_out) *nparts_out = a.ngrp;
  static const bool ok1 = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, false, true>));
  static const bool ok2 = big_lds_ok(reinterpret_cast<const void*>(&k_big_gemm<true, true, true>));
  if (!ok1 || !ok2) return 0;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  if (dy->a) edet_launch(k_big_gemm<true, true, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_gemm<true, false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_conv_bwd_data(big)");
  return 1;
}

// weight gradient: dweight fp32 HWIO [k][k][cin][cout] += gathered(in)^T dy
int pwb_try_conv_wgrad(const edet_tview_t* in, const edet_gview_t* dy, int k, int s, float* dweight, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
  if (in->act > EDET_ACT_SWISH) return 0;     // relu / relu6 / hswish: the direct dense-convolution kernels (conv.hip)
  using namespace pwb;
  const int cin = in->c, N = dy->c, K = k * k * cin;
  if (!workspace || cin % 8 != 0 || N % 8 != 0 || dy->ld % 8 != 0 || in->ld % 8 != 0) return 0;
  WgArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy; a.ws = reinterpret_cast<float*>(workspace);
  a.M = dy->n * dy->h * dy->w; a.K = K; a.N = N; a.hw = dy->h * dy->w;
  a.ck = k; a.cs = s; a.cin = cin; a.ih = in->h; a.iw = in->w; a.cow = dy->w; a.cohw = dy->h * dy->w;
  a.pad_t = same_pad_before(in->h, k, s); a.pad_l = same_pad_before(in->w, k, s);
  a.ntk = (K + 127) / 128; a.ntn = (N + 127) / 128;
  const int ntile = a.ntk * a.ntn;
  const int64_t kn = (int64_t)K * N;
  int S = (2048 + ntile - 1) / ntile;
  const int max_by_rows = (a.M + 4 * BK - 1) / (4 * BK);
  if (S > max_by_rows) S = max_by_rows;
  const int64_t max_by_ws = (int64_t)(workspace_bytes / sizeof(float)) / kn;
  if (S > max_by_ws) S = (int)max_by_ws;
  if (S < 1) return 0;
  a.rows_per_split = ((a.M + S - 1) / S + BK - 1) / BK * BK;
  a.S = (a.M + a.rows_per_split - 1) / a.rows_per_split;
  static const bool ok1 = big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad<false, true>));
  static const bool ok2 = big_lds_ok(reinterpret_cast<const void*>(&k_big_wgrad<true, true>));
  if (!ok1 || !ok2) return 0;
  const int grid = (a.S + 7) / 8 * 8 * ntile;
  if (dy->a) edet_launch(k_big_wgrad<true, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  else edet_launch(k_big_wgrad<false, true>, dim3(grid), dim3(THREADS), SMEM_BYTES, st, a);
  EDET_LAUNCH_CHECK("edet_conv_bwd_weight(big)");
  if (edet_reduce_partials(a.ws, a.S, kn, dweight, st) != 0) return -2;
  return 1;
}
