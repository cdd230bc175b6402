// Stem: dense 3x3 stride-2 convolution with Cin = 3 (TF 'SAME'), plus parameter cast kernels.
//
// Cin = 3 gives K = 27: too thin for the matrix cores and only ~3 % of the network's bytes, so
// this is a direct VALU convolution: the input tile is staged in LDS (fp32), each thread owns one
// output pixel and all Cout channels, weights are broadcast-read from LDS, the output row segment
// written by a wave is contiguous (Cout * 2 B per lane).
// Reference: efficientdet/backbone/efficientnet_model.py:506-527 (Stem), :511-519 (Conv2D).
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int TH = 8, TW = 32;  // output tile, one pixel per thread
constexpr int IH = (TH - 1) * 2 + 3, IW = (TW - 1) * 2 + 3;

struct StemArgs {
  const void* img;  // [n,h,w,3]
  int n, h, w, oh, ow, pad_t, pad_l, cout;
  const float* wgt;  // [3][3][3][cout]
  void* out; int ldo;
  float* stat_partials;
  edet_gview_t gy;
  float* dweight;
  float* ws;        // weight gradient: per-workgroup partials [P][27 * cout] (NULL: one workgroup, which adds into dweight)
  int tiles_y, tiles_x, nsp, P;
};

template <typename T>
__device__ __forceinline__ void stage_image_tile(const StemArgs& a, int n, int iy0, int ix0, float* tile) {
  // tile [IH][IW*3]; rows are contiguous runs of IW*3 elements in memory
  for (int q = threadIdx.x; q < IH * IW * 3; q += THREADS) {
    const int ly = q / (IW * 3), r = q - ly * (IW * 3);
    const int lx = r / 3, ci = r - lx * 3;
    const int gy = iy0 + ly, gx = ix0 + lx;
    float v = 0.f;
    if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
      v = to_f<T>(reinterpret_cast<const T*>(a.img)[((size_t)(n * a.h + gy) * a.w + gx) * 3 + ci]);
    tile[q] = v;
  }
}

// ---- bf16 forward on the matrix cores ------------------------------------------------------------
// im2col GEMM with K = 27 (padded to 32), computed transposed so that every lane ends up with
// contiguous channels of ONE pixel:  D'[channel][pixel] = sum_k W[k][channel] * patch[pixel][k].
//   A' fragment (weights):  lane l -> channel (l & 31), k = 8*(l>>5) + 16*ks + e      (registers, loaded once)
//   B' fragment (im2col):   lane l -> pixel   (l & 31), same k; k = ky*9 + r indexes 9 contiguous
//                           input elements (3 pixels x 3 channels) of input row 2*oy + ky.
// Workgroup tile = 2 output rows x 64 pixels (wave w: row w>>1, pixels (w&1)*32..+32); the 5 x 129
// pixel input patch is staged once in LDS as raw bf16.  HBM traffic = image read once + output written
// once (16-byte chunks per lane); BatchNorm statistic partials come from the rounded stored values.
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int MPX = 64;                      // pixels per tile row
constexpr int MROWS = 2;                     // output rows per tile
constexpr int MIW = (MPX - 1) * 2 + 3;       // 129 input pixels
constexpr int MIH = (MROWS - 1) * 2 + 3;     // 5 input rows
constexpr int MROWP = MIW * 3 + 5;           // 392 elements per LDS row
constexpr int MZERO = MIH * MROWP;           // index of an always-zero element

constexpr int MCH = MROWP / 8;                // 49 16-byte chunks per LDS row (387 elements used)
static_assert(MCH * 8 == MROWP && MIH * MCH <= THREADS, "one staging chunk per thread");

// Image staging, fast path (r06): the tile's five input rows as 16-byte chunks, ONE per thread, instead of 1935 two-byte
// loads (7.6 per thread, 128 bytes per wave instruction) -- legal when a tile row starts on a 16-byte boundary and chunks
// never straddle the right image edge: pad_l == 0 and w % 8 == 0 (every even size; otherwise the element loop).
// Chunk `tid` of tile `sp` (zeros outside the image, for sp beyond the last tile and for tid >= 245).
__device__ __forceinline__ bool stem_fast(const StemArgs& a) { return a.pad_l == 0 && (a.w & 7) == 0; }
__device__ __forceinline__ uint4 stem_chunk(const StemArgs& a, const bf16_t* img, int sp, int tid) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (sp < a.nsp && tid < MIH * MCH) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = sp / per_img, rr = sp - n * per_img;
    const int oy0 = (rr / a.tiles_x) * MROWS, ox0 = (rr % a.tiles_x) * MPX;
    const int gy = oy0 * 2 - a.pad_t + tid / MCH, gx3 = ox0 * 6 + (tid % MCH) * 8;
    if (gy >= 0 && gy < a.h && gx3 < a.w * 3)
      v = *reinterpret_cast<const uint4*>(img + (size_t)(n * a.h + gy) * a.w * 3 + gx3);
  }
  return v;
}

template <int NCT>  // channel tiles of 32
__global__ __launch_bounds__(THREADS) void k_stem_fwd_mfma(const StemArgs a) {
  __shared__ __align__(16) bf16_t tile[MIH * MROWP + 8];
  __shared__ float red[2 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int cout = a.cout;
  const bool want_stats = a.stat_partials != nullptr;
  const bf16_t* img = reinterpret_cast<const bf16_t*>(a.img);
  bf16_t* out = reinterpret_cast<bf16_t*>(a.out);

  // weight fragments and the 16 LDS gather offsets of this lane
  bf16x8 wf[NCT][2];
  int goff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * h + 16 * ks + e;
      const int ky = k / 9, r = k - ky * 9;
      goff[ks][e] = k < 27 ? ky * MROWP + r : -1;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int ch = ct * 32 + j;
        const float w = (k < 27 && ch < cout) ? a.wgt[k * cout + ch] : 0.f;
        wf[ct][ks][e] = (__bf16)w;
      }
    }
  float s1[NCT][16], s2[NCT][16];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int t = 0; t < 16; ++t) s1[ct][t] = s2[ct][t] = 0.f;
  for (int i = tid; i < 2 * 64; i += THREADS) red[i] = 0.f;
  if (tid < 8) tile[MZERO + tid] = 0;
  const bool fast = stem_fast(a);
  uint4 nxt = fast ? stem_chunk(a, img, blockIdx.x, tid) : make_uint4(0, 0, 0, 0);

  const int wr = wave >> 1, wx = (wave & 1) * 32 + j;   // this lane's pixel inside the tile
  for (int sp = blockIdx.x; sp < a.nsp; sp += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = sp / per_img, rr = sp - n * per_img;
    const int oy0 = (rr / a.tiles_x) * MROWS, ox0 = (rr % a.tiles_x) * MPX;
    const int iy0 = oy0 * 2 - a.pad_t, ix3 = (ox0 * 2 - a.pad_l) * 3;
    __syncthreads();
    if (fast) {
      // one 16-byte chunk per thread, loaded while the previous tile was on the matrix cores
      if (tid < MIH * MCH) *reinterpret_cast<uint4*>(tile + (tid / MCH) * MROWP + (tid % MCH) * 8) = nxt;
      nxt = stem_chunk(a, img, sp + a.P, tid);
    } else {
      for (int q = tid; q < MIH * MIW * 3; q += THREADS) {
        const int ly = q / (MIW * 3), r = q - ly * (MIW * 3);
        const int gy = iy0 + ly, gx3 = ix3 + r;
        bf16_t v = 0;
        if (gy >= 0 && gy < a.h && gx3 >= 0 && gx3 < a.w * 3) v = img[(size_t)(n * a.h + gy) * a.w * 3 + gx3];
        tile[ly * MROWP + r] = v;
      }
    }
    __syncthreads();
    const int base = (2 * wr) * MROWP + wx * 6;
    bf16x8 bfrag[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 raw;
#pragma unroll
      for (int e = 0; e < 8; ++e) raw[e] = (short)tile[goff[ks][e] >= 0 ? base + goff[ks][e] : MZERO];
      bfrag[ks] = __builtin_bit_cast(bf16x8, raw);
    }
    const int oy = oy0 + wr, ox = ox0 + wx;
    const bool pix_ok = oy < a.oh && ox < a.ow;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ct][0], bfrag[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ct][1], bfrag[1], acc, 0, 0, 0);
      // lane (pixel j, half h) holds channels (t&3) + 8*(t>>2) + 4*h.  Exchange 4-channel groups with the
      // other half so that each lane owns two runs of 8 contiguous channels:
      //   h = 0: ch 0-7 and 16-23;  h = 1: ch 8-15 and 24-31   (relative to ct*32)
      float v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = acc[t];
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lo_reg = half * 8 + 4 + i, hi_reg = half * 8 + i;
          const float give = h == 0 ? v[lo_reg] : v[hi_reg];
          const float got = __shfl_xor(give, 32, 64);
          if (h == 0) v[lo_reg] = got; else v[hi_reg] = got;
        }
#pragma unroll
      for (int run = 0; run < 2; ++run) {
        const int ch0 = ct * 32 + run * 16 + h * 8;
        if (pix_ok && ch0 < cout) {
          uint4 pk;
          pk.x = pack2bf(v[run * 8 + 0], v[run * 8 + 1]);
          pk.y = pack2bf(v[run * 8 + 2], v[run * 8 + 3]);
          pk.z = pack2bf(v[run * 8 + 4], v[run * 8 + 5]);
          pk.w = pack2bf(v[run * 8 + 6], v[run * 8 + 7]);
          *reinterpret_cast<uint4*>(out + ((size_t)(n * a.oh + oy) * a.ow + ox) * a.ldo + ch0) = pk;
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float r = bf2f(f2bf(v[run * 8 + e]));
              s1[ct][run * 8 + e] += r;
              s2[ct][run * 8 + e] += r * r;
            }
          }
        }
      }
    }
  }
  if (want_stats) {
    // reduce over the 32 pixel lanes of each half, then over the 4 waves through LDS
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float u = s1[ct][t], w = s2[ct][t];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { u += __shfl_down(u, off, 64); w += __shfl_down(w, off, 64); }
        s1[ct][t] = u;
        s2[ct][t] = w;
      }
    // the four waves add their sums in wave order (r04: no LDS atomics -- the same statistics on every run)
    for (int wv = 0; wv < THREADS / 64; ++wv) {
      if (wave == wv && j == 0) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int ch = ct * 32 + (t >> 3) * 16 + h * 8 + (t & 7);
            if (ch < cout) {
              red[ch] = (wv == 0 ? 0.f : red[ch]) + s1[ct][t];
              red[64 + ch] = (wv == 0 ? 0.f : red[64 + ch]) + s2[ct][t];
            }
          }
      }
      __syncthreads();
    }
    for (int i = tid; i < 2 * cout; i += THREADS) {
      const int which = i / cout, c = i - which * cout;
      a.stat_partials[(size_t)blockIdx.x * 2 * cout + i] = red[which * 64 + c];
    }
  }
}

// ---- bf16 weight gradient on the matrix cores (r03t) ---------------------------------------------------
// dW[k][ch] = sum over the output pixels of patch[pix][k] * dy[pix][ch]: the same im2col view as the forward, with the
// PIXEL as the reduction index of a 32x32x16 MFMA, D[i = k][j = channel]:
//   A fragment (patches): lane l -> k = l & 31, pixels 8*(l>>5) + 16*ks + e of the wave's 32: eight 2-byte gathers from
//                         the raw image tile (stride = 2 input pixels = 6 elements);
//   B fragment (dy):      lane l -> channel l & 31, the same eight pixels: dy = a*dz + b*y + c is staged per wave as a
//                         row-major [32 pixels][channels] bf16 tile and read through the LDS transpose read.
// The VALU kernel below (one fp32 multiply-add per (tap, channel, pixel): 864 per pixel) ran at 1.0 TB/s on the
// 320x320x32 gradient of D0 640x640 (1.09 ms); here a 16-pixel k-step costs 8 gathers, 2 transpose reads and one MFMA.
// dy is rounded to bf16 for the matrix cores, as in every other weight-gradient kernel of the bf16 path.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v_t;
template <int NCT>  // channel tiles of 32
__global__ __launch_bounds__(THREADS) void k_stem_bwd_weight_mfma(const StemArgs a) {
  constexpr int CS = NCT * 4;                        // 16-byte channel chunks per dy row (slots; cout / 8 of them are live)
  constexpr int PP = 64 / CS;                        // pixels per staging pass
  constexpr int DROW = NCT * 64 + 16;                // bytes per dy row (+ one 16-byte slot: conflict-free transpose reads)
  __shared__ __align__(16) bf16_t tile[MIH * MROWP + 8];
  __shared__ __align__(16) unsigned char dyt[4][32 * DROW];
  __shared__ float red[27 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int cout = a.cout;
  const bf16_t* img = reinterpret_cast<const bf16_t*>(a.img);
  const bf16_t* DZ = reinterpret_cast<const bf16_t*>(a.gy.dz);
  const bf16_t* YY = reinterpret_cast<const bf16_t*>(a.gy.y);
  const bool gbn = a.gy.a != nullptr;
  const int goff = i < 27 ? (i / 9) * MROWP + (i % 9) : -1;
  // dy staging: lane -> (pixel lane / CS of a pass, chunk lane % CS); the BatchNorm backward coefficients of the chunk
  const int sc = lane % CS, sp0 = lane / CS;
  const bool c_ok = sc * 8 < cout;
  float ga[8], gb[8], gc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] = 1.f; gb[e] = 0.f; gc[e] = 0.f; }
  if (gbn && c_ok) { loadf8(a.gy.a + sc * 8, ga); loadf8(a.gy.b + sc * 8, gb); loadf8(a.gy.cc + sc * 8, gc); }
  for (int q = tid; q < 27 * 64; q += THREADS) red[q] = 0.f;
  if (tid < 8) tile[MZERO + tid] = 0;
  const bool fast = stem_fast(a);
  uint4 nxt = fast ? stem_chunk(a, img, blockIdx.x, tid) : make_uint4(0, 0, 0, 0);
  f32x16 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[ct][t] = 0.f;
  unsigned char* my = dyt[wave];
  const int wr = wave >> 1, wx0 = (wave & 1) * 32;      // this wave's output row and first pixel inside the tile
  const int fi = lane & 15, fg = lane >> 4;             // transpose-read coordinates: 16-lane group fg = (half, column block)

  for (int sp = blockIdx.x; sp < a.nsp; sp += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = sp / per_img, rr = sp - n * per_img;
    const int oy0 = (rr / a.tiles_x) * MROWS, ox0 = (rr % a.tiles_x) * MPX;
    const int iy0 = oy0 * 2 - a.pad_t, ix3 = (ox0 * 2 - a.pad_l) * 3;
    __syncthreads();
    if (fast) {
      // one 16-byte chunk per thread, loaded while the previous tile was on the matrix cores
      if (tid < MIH * MCH) *reinterpret_cast<uint4*>(tile + (tid / MCH) * MROWP + (tid % MCH) * 8) = nxt;
      nxt = stem_chunk(a, img, sp + a.P, tid);
    } else {
      for (int q = tid; q < MIH * MIW * 3; q += THREADS) {
        const int ly = q / (MIW * 3), r = q - ly * (MIW * 3);
        const int gy = iy0 + ly, gx3 = ix3 + r;
        bf16_t v = 0;
        if (gy >= 0 && gy < a.h && gx3 >= 0 && gx3 < a.w * 3) v = img[(size_t)(n * a.h + gy) * a.w * 3 + gx3];
        tile[ly * MROWP + r] = v;
      }
    }
    // dy rows of this wave's 32 pixels -> its LDS tile (zero outside the image and beyond cout)
    const int oy = oy0 + wr;
#pragma unroll
    for (int ps = 0; ps < 32 / PP; ++ps) {
      const int p = ps * PP + sp0;
      const int ox = ox0 + wx0 + p;
      uint4 pk = make_uint4(0, 0, 0, 0);
      if (c_ok && oy < a.oh && ox < a.ow) {
        const size_t off = ((size_t)(n * a.oh + oy) * a.ow + ox) * a.gy.ld + sc * 8;
        float g[8];
        load8<bf16_t>(DZ + off, g);
        if (gbn) {
          float y[8];
          load8<bf16_t>(YY + off, y);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = fmaf(ga[e], g[e], fmaf(gb[e], y[e], gc[e]));
        }
        pk.x = pack2bf(g[0], g[1]); pk.y = pack2bf(g[2], g[3]); pk.z = pack2bf(g[4], g[5]); pk.w = pack2bf(g[6], g[7]);
      }
      *reinterpret_cast<uint4*>(my + p * DROW + sc * 16) = pk;
    }
    __syncthreads();
    const int base = (2 * wr) * MROWP + wx0 * 6;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 raw;
#pragma unroll
      for (int e = 0; e < 8; ++e) raw[e] = (short)tile[goff >= 0 ? base + (16 * ks + 8 * h + e) * 6 + goff : MZERO];
      const bf16x8 af = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        // 16-lane group fg: pixels 16 ks + 8 (fg >> 1) + 0..7, channels ct*32 + 16 (fg & 1) + 0..15; lane fi supplies the
        // address of 4 channels of pixel row fi / 4 and receives the 4 (then the next 4) pixels of channel fi
        const unsigned char* pr = my + (size_t)(16 * ks + 8 * (fg >> 1) + (fi >> 2)) * DROW + (ct * 32 + 16 * (fg & 1) + (fi & 3) * 4) * 2;
        typedef __attribute__((address_space(3))) bf16x4v_t* lds_ptr_t;
        const bf16x4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(pr));
        const bf16x4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(pr + 4 * DROW));
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[ct], 0, 0, 0);
      }
    }
  }
  // lane (channel j = lane & 31, half h) holds k = (t & 3) + 8 (t >> 2) + 4 h: the four waves are combined in LDS
  // (r04: in wave order, no LDS atomics; with a workspace the workgroup's partial goes there and edet_reduce_partials
  // adds the partials in order -- the same gradient on every run)
  for (int wv = 0; wv < THREADS / 64; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int k = (t & 3) + 8 * (t >> 2) + 4 * h, ch = ct * 32 + i;
          if (k < 27 && ch < cout) red[k * 64 + ch] += acc[ct][t];
        }
    }
  }
  __syncthreads();
  for (int q = tid; q < 27 * cout; q += THREADS) {
    const int k = q / cout, ch = q - k * cout;
    if (a.ws) a.ws[(size_t)blockIdx.x * 27 * cout + q] = red[k * 64 + ch];
    else a.dweight[q] += red[k * 64 + ch];        // (no workspace: launched as ONE workgroup, see stem_launch)
  }
}

// ---- fp32 (validation) forward: one output pixel per thread, direct VALU convolution --------------
template <typename T, int CV>  // CV = cout / 8
__global__ __launch_bounds__(THREADS) void k_stem_fwd(const StemArgs a) {
  constexpr int CO = CV * 8;
  __shared__ __align__(16) float tile[IH * IW * 3];
  __shared__ __align__(16) float wl[27 * CO];
  __shared__ float red[2 * CO];
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * CO; i += THREADS) wl[i] = a.wgt[i];
  for (int i = tid; i < 2 * CO; i += THREADS) red[i] = 0.f;
  const int ty = tid / TW, tx = tid % TW;
  const bool want_stats = a.stat_partials != nullptr;
  float s1[CO], s2[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) s1[c] = s2[c] = 0.f;

  for (int sp = blockIdx.x; sp < a.nsp; sp += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = sp / per_img, r = sp - n * per_img;
    const int oy0 = (r / a.tiles_x) * TH, ox0 = (r % a.tiles_x) * TW;
    __syncthreads();
    stage_image_tile<T>(a, n, oy0 * 2 - a.pad_t, ox0 * 2 - a.pad_l, tile);
    __syncthreads();
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < a.oh && ox < a.ow) {
      float acc[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            const float x = tile[(ty * 2 + ky) * (IW * 3) + (tx * 2 + kx) * 3 + ci];
            const float* wr = &wl[((ky * 3 + kx) * 3 + ci) * CO];
#pragma unroll
            for (int c = 0; c < CO; c += 4) {
              const float4 w4 = *reinterpret_cast<const float4*>(wr + c);
              acc[c] = fmaf(x, w4.x, acc[c]);
              acc[c + 1] = fmaf(x, w4.y, acc[c + 1]);
              acc[c + 2] = fmaf(x, w4.z, acc[c + 2]);
              acc[c + 3] = fmaf(x, w4.w, acc[c + 3]);
            }
          }
      T* dst = reinterpret_cast<T*>(a.out) + ((size_t)(n * a.oh + oy) * a.ow + ox) * a.ldo;
#pragma unroll
      for (int v = 0; v < CV; ++v) store8<T>(dst + v * 8, &acc[v * 8]);
      if (want_stats) {
#pragma unroll
        for (int c = 0; c < CO; ++c) { s1[c] += acc[c]; s2[c] += acc[c] * acc[c]; }
      }
    }
  }
  if (want_stats) {
    // wave shuffles, then the waves one after the other (a fixed order, no LDS atomics: the same partial row on every run)
#pragma unroll
    for (int c = 0; c < CO; ++c) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { s1[c] += __shfl_down(s1[c], off, 64); s2[c] += __shfl_down(s2[c], off, 64); }
    }
    __syncthreads();
    for (int wv = 0; wv < THREADS / 64; ++wv) {
      if (tid == wv * 64) {
#pragma unroll
        for (int c = 0; c < CO; ++c) { red[c] += s1[c]; red[CO + c] += s2[c]; }
      }
      __syncthreads();
    }
    for (int i = tid; i < 2 * CO; i += THREADS)
      a.stat_partials[(size_t)blockIdx.x * 2 * CO + i] = red[i];
  }
}

// dW[tap][co] = sum_pixels x[pixel, tap] * dy[pixel, co]; tap = (ky*3+kx)*3+ci.
// Thread = (pixel slice s, window position ky*3+kx, channel quad q): per pixel it reads the 3 input channels
// of its window position (LDS broadcast across the quads) and one float4 of dy, and does 12 FMAs into
// register accumulators -- 1/3 LDS instruction per FMA (the tap-major version needed 5/4).  The slices are
// combined through LDS one after the other and the workgroup's sums go to its partial row of the workspace
// (edet_reduce_partials adds the rows in order) or, for a single workgroup, into dW: no atomics, the same bits on every run.
template <typename T, int CV>
__global__ __launch_bounds__(THREADS) void k_stem_bwd_weight(const StemArgs a) {
  constexpr int CO = CV * 8;
  constexpr int NQ = CO / 4;                  // channel quads
  constexpr int S = THREADS / (9 * NQ);       // pixel slices
  static_assert(S >= 1, "cout too large for one workgroup");
  constexpr int HP = TH * TW / 2;             // pixels per half tile
  __shared__ __align__(16) float tile[IH * IW * 3];
  __shared__ __align__(16) float dyt[HP * CO];
  const int tid = threadIdx.x;
  const int q = tid % NQ, kk = (tid / NQ) % 9, sl = tid / (9 * NQ);
  const int ky = kk / 3, kx = kk - ky * 3;
  const bool active = sl < S;
  float acc[3][4];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[ci][e] = 0.f;

  for (int sp = blockIdx.x; sp < a.nsp; sp += a.P) {
    const int per_img = a.tiles_y * a.tiles_x;
    const int n = sp / per_img, r = sp - n * per_img;
    const int oy0 = (r / a.tiles_x) * TH, ox0 = (r % a.tiles_x) * TW;
    __syncthreads();
    stage_image_tile<T>(a, n, oy0 * 2 - a.pad_t, ox0 * 2 - a.pad_l, tile);
    for (int half = 0; half < 2; ++half) {
      if (half) __syncthreads();
      for (int i = tid; i < HP * CV; i += THREADS) {
        const int v = i % CV, lp = i / CV, pix = half * HP + lp;
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
        float g[8];
        if (oy < a.oh && ox < a.ow) {
          GradCoef gc;
          grad_load_coef(a.gy, v * 8, gc);
          grad_load<T>(a.gy, gc, ((size_t)(n * a.oh + oy) * a.ow + ox) * a.gy.ld + v * 8, g);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = 0.f;
        }
        store8<float>(&dyt[lp * CO + v * 8], g);
      }
      __syncthreads();
      if (active) {
#pragma unroll 4
        for (int lp = sl; lp < HP; lp += S) {
          const float4 g = *reinterpret_cast<const float4*>(&dyt[lp * CO + q * 4]);
          const int pix = half * HP + lp;
          const int ty = pix / TW, tx = pix % TW;
          const float* xp = &tile[(ty * 2 + ky) * (IW * 3) + (tx * 2 + kx) * 3];
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            const float x = xp[ci];
            acc[ci][0] = fmaf(x, g.x, acc[ci][0]);
            acc[ci][1] = fmaf(x, g.y, acc[ci][1]);
            acc[ci][2] = fmaf(x, g.z, acc[ci][2]);
            acc[ci][3] = fmaf(x, g.w, acc[ci][3]);
          }
        }
      }
    }
  }
  // combine the pixel slices: dyt is free after the last barrier-separated use
  __syncthreads();
  float* red = dyt;                           // [27][CO]
  for (int i = tid; i < 27 * CO; i += THREADS) red[i] = 0.f;
  __syncthreads();
  for (int sv = 0; sv < S; ++sv) {
    if (active && sl == sv) {
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(kk * 3 + ci) * CO + q * 4 + e] += acc[ci][e];
    }
    __syncthreads();
  }
  for (int i = tid; i < 27 * CO; i += THREADS) {
    if (a.ws) a.ws[(size_t)blockIdx.x * 27 * CO + i] = red[i];
    else a.dweight[i] += red[i];
  }
}

template <typename T>
__global__ void k_cast(const float* __restrict__ src, T* __restrict__ dst, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f<T>(src[i]);
}

// dst[r][c] (ld_out, zero padded) = src[r][c]            (transpose == 0, dst rows = rows)
// dst[c][r] (ld_out, zero padded) = src[r][c]            (transpose == 1, dst rows = cols)
template <typename T>
__global__ void k_cast_matrix(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols,
                              int ld_out, int transpose) {
  const int drows = transpose ? cols : rows;
  const int dcols = transpose ? rows : cols;
  const int64_t total = (int64_t)drows * ld_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_out), c = (int)(i - (int64_t)r * ld_out);
    float v = 0.f;
    if (c < dcols) v = transpose ? src[(size_t)c * cols + r] : src[(size_t)r * cols + c];
    dst[i] = from_f<T>(v);
  }
}

// every item of a cast plan in one launch: blockIdx.y = item, blockIdx.x strides over its elements
template <typename T>
__global__ void k_cast_batch(const edet_cast_item_t* __restrict__ items) {
  const edet_cast_item_t it = items[blockIdx.y];
  const float* src = it.src;
  T* dst = reinterpret_cast<T*>(it.dst);
  const int drows = it.transpose ? it.cols : it.rows;
  const int dcols = it.transpose ? it.rows : it.cols;
  const int64_t total = (int64_t)drows * it.ld_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / it.ld_out), c = (int)(i - (int64_t)r * it.ld_out);
    float v = 0.f;
    if (c < dcols) v = it.transpose ? src[(size_t)c * it.cols + r] : src[(size_t)r * it.cols + c];
    dst[i] = from_f<T>(v);
  }
}

template <typename T>
int stem_launch(bool fwd, StemArgs& a, hipStream_t st) {
  if (!fwd && sizeof(T) == 2) {
    a.tiles_y = cdiv(a.oh, MROWS);
    a.tiles_x = cdiv(a.ow, MPX);
    a.nsp = a.n * a.tiles_y * a.tiles_x;
    a.P = a.nsp < EDET_MAX_PARTS ? a.nsp : EDET_MAX_PARTS;
    if (!a.ws) a.P = 1;
    EDET_CHECK(a.cout <= 64, "stem: cout %d unsupported (need <= 64)", a.cout);
    if (a.cout <= 32) edet_launch(k_stem_bwd_weight_mfma<1>, dim3(a.P), dim3(THREADS), 0, st, a);
    else edet_launch(k_stem_bwd_weight_mfma<2>, dim3(a.P), dim3(THREADS), 0, st, a);
    EDET_LAUNCH_CHECK("edet_stem_bwd_weight");
    if (a.ws && edet_reduce_partials(a.ws, a.P, (int64_t)27 * a.cout, a.dweight, st) != 0) return -2;
    return 0;
  }
  if (fwd && sizeof(T) == 2) {
    a.tiles_y = cdiv(a.oh, MROWS);
    a.tiles_x = cdiv(a.ow, MPX);
    a.nsp = a.n * a.tiles_y * a.tiles_x;
    a.P = a.nsp < EDET_MAX_PARTS ? a.nsp : EDET_MAX_PARTS;
    EDET_CHECK(a.cout <= 64, "stem: cout %d unsupported (need <= 64)", a.cout);
    if (a.cout <= 32) edet_launch(k_stem_fwd_mfma<1>, dim3(a.P), dim3(THREADS), 0, st, a);
    else edet_launch(k_stem_fwd_mfma<2>, dim3(a.P), dim3(THREADS), 0, st, a);
    EDET_LAUNCH_CHECK("edet_stem_fwd");
    return 0;
  }
  a.tiles_y = cdiv(a.oh, TH);
  a.tiles_x = cdiv(a.ow, TW);
  a.nsp = a.n * a.tiles_y * a.tiles_x;
  a.P = a.nsp < EDET_MAX_PARTS ? a.nsp : EDET_MAX_PARTS;
  if (!fwd && !a.ws) a.P = 1;       // no workspace for the partial rows: one workgroup adds into dW directly
  const dim3 grid(a.P), block(THREADS);
#define STEM_CASE(CV)                                                   \
  case CV:                                                              \
    if (fwd) {                                                          \
      if constexpr (sizeof(T) == 4) edet_launch(k_stem_fwd<T, CV>, grid, block, 0, st, a); \
    } else {                                                            \
      edet_launch(k_stem_bwd_weight<T, CV>, grid, block, 0, st, a);              \
    }                                                                   \
    break;
  switch (a.cout / 8) {
    STEM_CASE(3) STEM_CASE(4) STEM_CASE(5) STEM_CASE(6) STEM_CASE(7) STEM_CASE(8)
    default:
      EDET_CHECK(false, "stem: cout %d unsupported (need 24,32,40,48,56,64)", a.cout);
  }
#undef STEM_CASE
  EDET_LAUNCH_CHECK("edet_stem");
  if (!fwd && a.ws && edet_reduce_partials(a.ws, a.P, (int64_t)27 * a.cout, a.dweight, st) != 0) return -2;
  return 0;
}

void stem_geometry(StemArgs& a) {
  a.oh = same_out(a.h, 2);
  a.ow = same_out(a.w, 2);
  a.pad_t = same_pad_before(a.h, 3, 2);
  a.pad_l = same_pad_before(a.w, 3, 2);
}

}  // namespace

extern "C" int edet_stem_fwd(const void* images, int n, int h, int w, const float* weight,
                             void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                             int dtype, void* stream) {
  EDET_CHECK(images && weight && out, "edet_stem_fwd: null pointer");
  EDET_CHECK(cout % 8 == 0 && ldo % 8 == 0 && ldo >= cout, "edet_stem_fwd: bad cout/ldo");
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.img = images; a.n = n; a.h = h; a.w = w; a.cout = cout; a.wgt = weight;
  a.out = out; a.ldo = ldo; a.stat_partials = stat_partials;
  stem_geometry(a);
  int rc = dtype == EDET_BF16 ? stem_launch<bf16_t>(true, a, to_stream(stream))
         : dtype == EDET_F32 ? stem_launch<float>(true, a, to_stream(stream)) : -1;
  if (rc == -1 && dtype != EDET_BF16 && dtype != EDET_F32) edet_set_error("edet_stem_fwd: bad dtype %d", dtype);
  if (nparts_out) *nparts_out = a.P;
  return rc;
}

extern "C" int edet_stem_bwd_weight(const void* images, int n, int h, int w,
                                    const edet_gview_t* dy, float* dweight, void* workspace, size_t workspace_bytes,
                                    int dtype, void* stream) {
  EDET_CHECK(images && dy && dy->dz && dweight, "edet_stem_bwd_weight: null pointer");
  EDET_CHECK(dy->c % 8 == 0 && dy->ld % 8 == 0, "edet_stem_bwd_weight: dy c/ld % 8");
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.img = images; a.n = n; a.h = h; a.w = w; a.cout = dy->c; a.gy = *dy; a.dweight = dweight;
  stem_geometry(a);
  EDET_CHECK(a.oh == dy->h && a.ow == dy->w, "edet_stem_bwd_weight: dy geometry mismatch");
  // ordered partial sums through the workspace when it holds EDET_MAX_PARTS of them (else: one workgroup, no partials)
  if (workspace && workspace_bytes >= (size_t)EDET_MAX_PARTS * 27 * dy->c * sizeof(float))
    a.ws = reinterpret_cast<float*>(workspace);
  if (dtype == EDET_BF16) return stem_launch<bf16_t>(false, a, to_stream(stream));
  if (dtype == EDET_F32) return stem_launch<float>(false, a, to_stream(stream));
  EDET_CHECK(false, "edet_stem_bwd_weight: bad dtype %d", dtype);
}

extern "C" int edet_cast(const float* src, void* dst, int64_t count, int dtype, void* stream) {
  EDET_CHECK(src && dst, "edet_cast: null pointer");
  if (count <= 0) return 0;
  const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  if (dtype == EDET_BF16) edet_launch(k_cast<bf16_t>, grid, dim3(256), 0, to_stream(stream), src, (bf16_t*)dst, count);
  else if (dtype == EDET_F32) edet_launch(k_cast<float>, grid, dim3(256), 0, to_stream(stream), src, (float*)dst, count);
  else EDET_CHECK(false, "edet_cast: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_cast");
  return 0;
}

extern "C" int edet_cast_matrix(const float* src, void* dst, int rows, int cols, int ld_out,
                                int transpose, int dtype, void* stream) {
  EDET_CHECK(src && dst, "edet_cast_matrix: null pointer");
  EDET_CHECK(ld_out >= (transpose ? rows : cols), "edet_cast_matrix: ld_out too small");
  const int64_t total = (int64_t)(transpose ? cols : rows) * ld_out;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  if (dtype == EDET_BF16)
    edet_launch(k_cast_matrix<bf16_t>, grid, dim3(256), 0, to_stream(stream), src, (bf16_t*)dst, rows, cols, ld_out, transpose);
  else if (dtype == EDET_F32)
    edet_launch(k_cast_matrix<float>, grid, dim3(256), 0, to_stream(stream), src, (float*)dst, rows, cols, ld_out, transpose);
  else EDET_CHECK(false, "edet_cast_matrix: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_cast_matrix");
  return 0;
}

extern "C" int edet_cast_batch(const edet_cast_item_t* items_dev, int count, int max_blocks_per_item, int dtype,
                               void* stream) {
  EDET_CHECK(items_dev && count > 0 && max_blocks_per_item > 0, "edet_cast_batch: bad arguments");
  const dim3 grid(max_blocks_per_item, count);
  if (dtype == EDET_BF16) edet_launch(k_cast_batch<bf16_t>, grid, dim3(256), 0, to_stream(stream), items_dev);
  else if (dtype == EDET_F32) edet_launch(k_cast_batch<float>, grid, dim3(256), 0, to_stream(stream), items_dev);
  else EDET_CHECK(false, "edet_cast_batch: bad dtype %d", dtype);
  EDET_LAUNCH_CHECK("edet_cast_batch");
  return 0;
}
