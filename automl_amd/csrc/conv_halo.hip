// Dense 3x3 convolution (TF 'SAME', stride 1 or 2), forward, from an LDS-resident halo tile.
//
// The implicit GEMM of pw_big.hip (k_big_gemm<.., CONV>) gathers, bounds-checks and transforms every input element once
// per TAP and per 128-column tile (9 x ntj times), with an integer division per 16-byte chunk: on the Fused-MBConv layers
// of EfficientNetV2-S (112x112x24 -> 24, 56x56x48 -> 192, 28x28x64 -> 256) it reaches 0.75-1.1 TB/s and 3-15 % of the MFMA
// peak (0.5 TB/s on the 16- and 32-channel layers of the b0-b3 / L variants).  Here a workgroup owns an 8 x 16 tile of output pixels (= the 128 rows of the MFMA tile):
//
//   * the 10 x 18 (stride 2: 17 x 33) input halo of the tile is loaded ONCE, transformed once (BatchNorm / activation of the producing layer;
//     'SAME' padding is zero in the activated domain) and parked in LDS, pixel-major, channels padded to a multiple of 16;
//   * the A fragments of tap (ky, kx) are plain ds_read_b128 at halo pixel (S py + ky, S px + kx): no im2col, no gather;
//   * the weights of one tap ([columns][cin], 6-16 KiB) stream through two LDS stages, requested one tap ahead; the (at
//     most two) 128-column tiles of the layer are walked one after the other on the SAME halo;
//   * 128-column tiles as 2 x 2 waves of 2 x 2 v_mfma_f32_32x32x16_bf16, or -- layers with at most 32 output channels --
//     32-column tiles as 4 x 1 waves of one MFMA tile each (no multiply-adds on 96 columns that do not exist);
//   * epilogue as k_big_gemm's forward: C tile through LDS, 16-byte stores, BatchNorm statistic partials in a fixed order
//     (a workgroup walks a contiguous range of tiles, one partial row per workgroup).
//
// Reference call sites replaced: tf.keras.layers.Conv2D k x k of FusedMBConvBlock, efficientnetv2/effnetv2_model.py:338-346
// (expand) and :362-371 (the single conv of expand_ratio 1 blocks); backbone/efficientnet_model.py has none (EfficientDet
// uses MBConv only).  Weights [cout][9 cin], reduction index (ky 3 + kx) cin + c contiguous (edet_conv_fwd's layout).
#include <stdlib.h>

#include "common.h"

namespace cvh {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int THREADS = 256;
constexpr int TH = 8, TW = 16, BM = TH * TW;     // output pixels per tile
// halo of a tile at stride S: (TH - 1) S + 3 rows, (TW - 1) S + 3 columns
constexpr int LDC_BF = 128 * 2 + 16;             // bf16 C tile row stride (272), BJ <= 128

struct Args {
  edet_tview_t tv;
  const bf16_t* Bm;   // [J][ldb]
  int ldb, J;
  bf16_t* out;
  int ldo;
  int oh, ow, pad_t, pad_l;          // output map, 'SAME' padding before
  int tiles_y, tiles_x, ntiles;      // per image / total
  int ntj, tpw, ngrp;
  float* stat_partials;
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

// CIN: input channels (multiple of 8).  WIDE: 128-column tiles (2 x 2 waves of 64 x 64), else 32-column tiles (4 x 1
// waves of 32 x 32).  S: stride (1, 2).
template <int CIN, bool WIDE, int S> struct Geo {
  static constexpr int HH = (TH - 1) * S + 3, HW = (TW - 1) * S + 3;
  static constexpr int KP = (CIN + 15) / 16 * 16;      // channels per tap in LDS (zero-padded)
  static constexpr int CPC = CIN / 8, CPP = KP / 8;    // 16-byte chunks per pixel: loaded / stored
  static constexpr int PS = KP * 2 + 16;               // halo pixel stride (bytes; +16: fragment reads of 32 pixels spread over the banks)
  static constexpr int HALO_BYTES = HH * HW * PS;
  static constexpr int BJ = WIDE ? 128 : 32;
  static constexpr int BS = KP * 2 + 16;               // weight row stride
  static constexpr int BSTAGE = BJ * BS;
  static constexpr int CBYTES = BM * (WIDE ? LDC_BF : 32 * 2 + 16);
  static constexpr int RED_BYTES = 2 * (THREADS / (BJ / 8)) * BJ * 4;      // statistics scratch [2][row lanes][BJ]
  static constexpr int R0 = 2 * BSTAGE > CBYTES ? 2 * BSTAGE : CBYTES;
  static constexpr int R = R0 > RED_BYTES ? R0 : RED_BYTES;                // weight stages, later the C tile / the sums
  static constexpr int HALO_PAD = (HALO_BYTES + 15) / 16 * 16;
  static constexpr int COEF_BYTES = 2 * KP * 4;                            // scale | shift of the input view
  static constexpr int SMEM_BYTES = HALO_PAD + R + COEF_BYTES;
  static constexpr int NHL = (HH * HW * CPP + THREADS - 1) / THREADS;      // halo chunks per thread
  static constexpr int BCH = BJ * CPP;                 // weight chunks per tap
  static constexpr int BPT = (BCH + THREADS - 1) / THREADS;
  static constexpr int WMT = WIDE ? 2 : 1, WJT = WIDE ? 2 : 1;     // MFMA tiles per wave
};

template <int CIN, bool WIDE, int S>
__global__ __launch_bounds__(THREADS, 2) void k_conv3_halo(const Args a) {
  using G = Geo<CIN, WIDE, S>;
  constexpr int HH = G::HH, HW = G::HW;
  constexpr int KP = G::KP, CPC = G::CPC, CPP = G::CPP, PS = G::PS, BJ = G::BJ, BS = G::BS, BPT = G::BPT;
  constexpr int WMT = G::WMT, WJT = G::WJT;
  constexpr int LDC = WIDE ? LDC_BF : 32 * 2 + 16;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* halo = smem;
  unsigned char* reg = smem + G::HALO_PAD;
  float* coef = reinterpret_cast<float*>(smem + G::HALO_PAD + G::R);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = WIDE ? (wave & 1) : wave, wj = WIDE ? (wave >> 1) : 0;
  const int grp = blockIdx.x;
  const int H = a.tv.h, W = a.tv.w;                    // input map
  const int OH = a.oh, OW = a.ow;                      // output map
  const bool want_stats = a.stat_partials != nullptr;
  const bool affine = a.tv.scale != nullptr;
  const int act = a.tv.act;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.tv.data);
  // epilogue geometry: thread -> 8 output columns ec*8.., rows er + ER*i
  constexpr int ECN = BJ / 8, ER = THREADS / ECN;        // 16 x 16 or 4 x 64
  const int ec = tid % ECN, er = tid / ECN;
  constexpr int MAXJT = WIDE ? 4 : 1;           // column tiles walked by a workgroup (cout <= 512 / 32: host check)
  float tot1[MAXJT], tot2[MAXJT];
#pragma unroll
  for (int t = 0; t < MAXJT; ++t) tot1[t] = tot2[t] = 0.f;
  for (int i = tid; i < 2 * KP; i += THREADS) {
    const int c = i < KP ? i : i - KP;
    coef[i] = affine && c < CIN ? (i < KP ? a.tv.scale[c] : a.tv.shift[c]) : (i < KP ? 1.f : 0.f);
  }

  // weight chunks of this thread: chunk q = tid + 256 i -> column q / CPP, channel chunk q % CPP (>= CPC: zero padding)
  const bf16_t* bsrc[BPT];
  int bdst[BPT];
  bool bok[BPT];
  auto b_setup = [&](int j0) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int q = tid + THREADS * i;
      const int col = q / CPP, cc = q - col * CPP;
      bok[i] = q < G::BCH && cc < CPC && j0 + col < a.J;
      bsrc[i] = a.Bm + (size_t)min(j0 + col, a.J - 1) * a.ldb + min(cc, CPC - 1) * 8;
      bdst[i] = q < G::BCH ? col * BS + cc * 16 : -1;
    }
  };
  // Weight prefetch: ND register sets, the set of tap t is requested ND taps before its MFMAs (the tap loop is unrolled, the
  // set index static).  r06ar: ND = 3 instead of 1 changes nothing (56x56x48->192 0.238 ms either way) at +25 VGPRs -- the
  // weight requests are not what a tile waits for
  constexpr int ND = 1;
  uint4 rb[ND][BPT];
  auto b_issue = [&](int tap, uint4 (&set)[BPT]) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      set[i] = make_uint4(0, 0, 0, 0);
      if (bok[i]) set[i] = *reinterpret_cast<const uint4*>(bsrc[i] + tap * CIN);
    }
  };
  auto b_commit = [&](unsigned char* stage, const uint4 (&set)[BPT]) {
#pragma unroll
    for (int i = 0; i < BPT; ++i)
      if (bdst[i] >= 0) *reinterpret_cast<uint4*>(stage + bdst[i]) = set[i];
  };

  const int per_img = a.tiles_y * a.tiles_x;
  const int t_end = min(a.ntiles, (grp + 1) * a.tpw);
  for (int tile = grp * a.tpw; tile < t_end; ++tile) {
    const int img = tile / per_img, rem = tile - img * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    __syncthreads();                        // the previous tile's epilogue is done with the LDS (first tile: coef is written)
    b_setup(0);
#pragma unroll
    for (int t = 0; t < ND; ++t) b_issue(t, rb[t]);
    // ---- halo: loaded (the requests of a pass of at most six chunks per thread first), transformed and parked once
    constexpr int NB = G::NHL < 6 ? G::NHL : 6;
#pragma unroll 1
    for (int base = 0; base < G::NHL; base += NB) {
      uint4 raw[NB];
      int hdst[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + THREADS * (base + i);
        const int p = e / CPP, c = e - p * CPP;
        const int hy = p / HW, hx = p - hy * HW;
        const int iy = oy0 * S + hy - a.pad_t, ix = ox0 * S + hx - a.pad_l;
        raw[i] = make_uint4(0, 0, 0, 0);
        const bool inside = e < HH * HW * CPP && c < CPC && iy >= 0 && iy < H && ix >= 0 && ix < W;
        // hdst: byte offset in the halo; bit 30: the chunk holds data (else zeros: padding pixel or padding channels)
        hdst[i] = e < HH * HW * CPP ? (p * PS + c * 16) | (inside ? 1 << 30 : 0) : -1;
        if (inside) raw[i] = *reinterpret_cast<const uint4*>(X + ((size_t)(img * H + iy) * W + ix) * a.tv.ld + c * 8);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (hdst[i] < 0) continue;
        uint4 v = raw[i];
        if ((hdst[i] >> 30) && (affine || act != EDET_ACT_NONE)) {
          const int c = ((tid + THREADS * (base + i)) % CPP) * 8;
          float x[8];
          unpack8(v, x);
          if (affine) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = fmaf(x[k], coef[c + k], coef[KP + c + k]);
          }
          if (act == EDET_ACT_SWISH) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = swishf_(x[k]);
          } else if (act != EDET_ACT_NONE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = act_other_(act, x[k]);
          }
          v = pack8(x);
        }
        *reinterpret_cast<uint4*>(halo + (hdst[i] & 0x3fffffff)) = v;
      }
    }
    // ---- the column tiles of this pixel tile, one after the other on the same halo
#pragma unroll
    for (int jt = 0; jt < MAXJT; ++jt) {
    if (jt < a.ntj) {
    const int j0 = jt * BJ;
    const int ej = j0 + ec * 8;
    const bool ecol_ok = ej < a.J;
    if (jt > 0) {
      __syncthreads();                      // the previous column tile's epilogue is done with the weight stages
      b_setup(j0);
#pragma unroll
      for (int t = 0; t < ND; ++t) b_issue(t, rb[t]);
    }
    b_commit(reg, rb[0]);
    b_issue(ND, rb[0]);
    __syncthreads();

    f32x16 acc[WJT][WMT];
#pragma unroll
    for (int x = 0; x < WJT; ++x)
#pragma unroll
      for (int y = 0; y < WMT; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    // halo pixel (top-left tap) of this lane's A rows
    int apix[WMT];
#pragma unroll
    for (int mi = 0; mi < WMT; ++mi) {
      const int m = (WIDE ? wm * 64 : wm * 32) + mi * 32 + r;
      apix[mi] = (m / TW) * S * HW + (m % TW) * S;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      unsigned char* cur = reg + (tap & 1) * G::BSTAGE;
      unsigned char* nxt = reg + ((tap + 1) & 1) * G::BSTAGE;
      if (tap + 1 < 9) b_commit(nxt, rb[(tap + 1) % ND]);
      if (tap + 1 + ND < 9) b_issue(tap + 1 + ND, rb[(tap + 1) % ND]);
      const int ky = tap / 3, kx = tap - ky * 3;
      const int toff = (ky * HW + kx) * PS;
#pragma unroll
      for (int kk = 0; kk < KP / 16; ++kk) {
        const int koff = (kk * 16 + h * 8) * 2;
        bf16x8 af[WMT], bfr[WJT];
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(halo + apix[mi] * PS + toff + koff);
#pragma unroll
        for (int nj = 0; nj < WJT; ++nj)
          bfr[nj] = *reinterpret_cast<const bf16x8*>(cur + ((WIDE ? wj * 64 : 0) + nj * 32 + r) * BS + koff);
#pragma unroll
        for (int nj = 0; nj < WJT; ++nj)
#pragma unroll
          for (int mi = 0; mi < WMT; ++mi)
            acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[nj], af[mi], acc[nj][mi], 0, 0, 0);
      }
      __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue through LDS
    // acc[nj][mi][4 g + e] = D[column 8 g + 4 h + e of the 32-tile][row r of the 32-tile]
#pragma unroll
    for (int nj = 0; nj < WJT; ++nj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = (WIDE ? wj * 64 : 0) + nj * 32 + 8 * g + 4 * h;
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) {
          uint2 pk;
          pk.x = pack2bf(acc[nj][mi][4 * g + 0], acc[nj][mi][4 * g + 1]);
          pk.y = pack2bf(acc[nj][mi][4 * g + 2], acc[nj][mi][4 * g + 3]);
          const int row = (WIDE ? wm * 64 : wm * 32) + mi * 32 + r;
          *reinterpret_cast<uint2*>(reg + row * LDC + ch * 2) = pk;
        }
      }
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < BM / ER; ++i) {
      const int row = er + ER * i;
      const int oy = oy0 + row / TW, ox = ox0 + row % TW;
      if (ecol_ok && oy < OH && ox < OW) {
        const uint4 v = *reinterpret_cast<const uint4*>(reg + row * LDC + ec * 16);
        *reinterpret_cast<uint4*>(a.out + ((size_t)(img * OH + oy) * OW + ox) * a.ldo + ej) = v;
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
      float* red = reinterpret_cast<float*>(reg);            // [2][ER][BJ]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[er * BJ + ec * 8 + e] = s1[e];
        red[(ER + er) * BJ + ec * 8 + e] = s2[e];
      }
      __syncthreads();
      if (tid < BJ) {
        for (int i = 0; i < ER; ++i) { tot1[jt] += red[i * BJ + tid]; tot2[jt] += red[(ER + i) * BJ + tid]; }
      }
    }
    }
    }
  }
  if (want_stats && tid < BJ) {
    float* dst = a.stat_partials + (size_t)grp * 2 * a.J;
#pragma unroll
    for (int jt = 0; jt < MAXJT; ++jt)
      if (jt < a.ntj && jt * BJ + tid < a.J) {
        dst[jt * BJ + tid] = tot1[jt];
        dst[a.J + jt * BJ + tid] = tot2[jt];
      }
  }
}

template <int CIN, bool WIDE, int S> int launch(const Args& a, hipStream_t st) {
  using G = Geo<CIN, WIDE, S>;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_halo<CIN, WIDE, S>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM_BYTES) == hipSuccess;
  if (!ok) return 0;
  edet_launch(k_conv3_halo<CIN, WIDE, S>, dim3(a.ngrp), dim3(THREADS), G::SMEM_BYTES, st, a);
  return 1;
}

}  // namespace cvh

// return 1 = handled, 0 = shape outside the envelope (the caller goes on to the implicit GEMM), < 0 = error.
// Envelope: 3 x 3, stride 1 (stride 2 up to 32 input channels), the Fused-MBConv widths of the EfficientNetV2 family (effnetv2_configs.py: 16 / 24 / 32 / 48 /
// 64 / 80 / 96 input channels), at most 512 output channels, no SE gate on the input view.
int cvh_try_conv_fwd(const edet_tview_t* in, const void* wt, int ldw, int k, int s, void* out, int cout, int ldo,
                     float* stat_partials, int* nparts_out, hipStream_t st) {
  using namespace cvh;
  if (k != 3 || (s != 1 && s != 2) || in->gate) return 0;
  const int cin = in->c;
  // every width measured faster than the implicit GEMM (r06at, batch 128: 16 / 32 channels 4.6-5x, 24: 4.1x, 48 / 64: 1.8-1.9x,
  // 80: 1.8x, 96: 1.2-1.6x)
  if (cin != 16 && cin != 24 && cin != 32 && cin != 48 && cin != 64 && cin != 80 && cin != 96) return 0;
  // stride 2: a 17 x 33 halo.  r06au (per launch, implicit GEMM -> halo tile): 16 / 24 / 32 channels 1.26-1.48x faster; 48
  // channels (63 KiB of halo, one workgroup per CU) 0.126 -> 0.138 ms: stays on the implicit GEMM
  if (s == 2 && cin > 32) return 0;
  if (in->ld % 8 != 0 || ldw % 8 != 0 || ldo % 8 != 0 || ldo < (cout + 7) / 8 * 8) return 0;
  const char* e = getenv("EDET_CONV_HALO");
  if (e && e[0] == '0') return 0;           // lab switch: the implicit GEMM for every shape
  Args a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = reinterpret_cast<const bf16_t*>(wt); a.ldb = ldw; a.J = cout;
  a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.oh = same_out(in->h, s); a.ow = same_out(in->w, s);
  a.pad_t = same_pad_before(in->h, 3, s); a.pad_l = same_pad_before(in->w, 3, s);
  a.tiles_y = (a.oh + TH - 1) / TH; a.tiles_x = (a.ow + TW - 1) / TW;
  a.ntiles = in->n * a.tiles_y * a.tiles_x;
  if (cout > 512) return 0;                 // a workgroup walks at most four 128-column tiles
  const bool wide = cout > 32;
  a.ntj = (cout + (wide ? 128 : 32) - 1) / (wide ? 128 : 32);
  a.tpw = (a.ntiles + EDET_MAX_PARTS - 1) / EDET_MAX_PARTS;
  a.ngrp = (a.ntiles + a.tpw - 1) / a.tpw;
  int rc = 0;
  if (s == 1) {
    switch (cin) {
      case 16: rc = wide ? launch<16, true, 1>(a, st) : launch<16, false, 1>(a, st); break;
      case 24: rc = wide ? launch<24, true, 1>(a, st) : launch<24, false, 1>(a, st); break;
      case 32: rc = wide ? launch<32, true, 1>(a, st) : launch<32, false, 1>(a, st); break;
      case 48: rc = wide ? launch<48, true, 1>(a, st) : launch<48, false, 1>(a, st); break;
      case 64: rc = wide ? launch<64, true, 1>(a, st) : launch<64, false, 1>(a, st); break;
      case 80: rc = wide ? launch<80, true, 1>(a, st) : launch<80, false, 1>(a, st); break;
      default: rc = wide ? launch<96, true, 1>(a, st) : launch<96, false, 1>(a, st); break;
    }
  } else {
    switch (cin) {
      case 16: rc = wide ? launch<16, true, 2>(a, st) : launch<16, false, 2>(a, st); break;
      case 24: rc = wide ? launch<24, true, 2>(a, st) : launch<24, false, 2>(a, st); break;
      default: rc = wide ? launch<32, true, 2>(a, st) : launch<32, false, 2>(a, st); break;
    }
  }
  if (rc != 1) return rc;
  if (nparts_out) *nparts_out = a.ngrp;
  EDET_LAUNCH_CHECK("edet_conv_fwd(halo)");
  return 1;
}
