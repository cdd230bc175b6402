// Pointwise (1x1) convolution on the gfx950 matrix cores.
//
//   forward : out[m, j]  = sum_k view(in)[m, k] * Wt[j, k] (+ bias[j])          (+ BN stat partials)
//   dgrad   : d in[m, k] = sum_j dy[m, j] * W[k, j]  -> chained through the input view's
//             activation / BatchNorm statistics / SE gate in the epilogue
//   wgrad   : dW[k, j]  += sum_m view(in)[m, k] * dy[m, j]
//
// All three are HBM-bound "tall-skinny" GEMMs (M = N*H*W up to 13 M rows, K and J <= 1152), so
// the design goal is: stream the big operand exactly once with 16-byte coalesced accesses, apply
// BatchNorm/swish/SE-gate (forward) or the BatchNorm backward (gradient) on load, keep the small
// operand in L2, and fuse the per-channel reductions into the epilogue.  MFMA 16x16x32 bf16
// (16x16x4 f32 in the fp32 validation mode) does the contraction; MFMA utilisation is not the
// limiter at these arithmetic intensities.
//
// Reference call sites: efficientdet/backbone/efficientnet_model.py:304-312,345-353;
// efficientdet/tf2/efficientdet_keras.py:195-207,286-290,459-464,546-556.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128;  // rows per workgroup tile
constexpr int BK = 32;   // reduction step
constexpr int THREADS = 256;

struct GemmArgs {
  edet_tview_t tv;  // fwd: A source.  bwd: the conv's *input* view (epilogue chain target)
  edet_gview_t gv;  // bwd: A source (dy)
  const void* Bm;   // [J][ldb], reduction index contiguous
  int ldb;
  int M, R, J, Jp;  // rows, reduction length, output columns, J rounded up to 8
  int hw;           // pixels per image
  int tiles_per_wg;
  const float* bias;  // fwd
  void* out;
  int ldo;
  edet_bwd_epi_t epi;  // bwd
  float* stat_partials;
};

template <typename T> struct Raw8;  // 8 raw elements in registers
template <> struct Raw8<bf16_t> { uint4 v; };
template <> struct Raw8<float> { float4 lo, hi; };

template <typename T> __device__ __forceinline__ void raw_zero(Raw8<T>& r);
template <> __device__ __forceinline__ void raw_zero<bf16_t>(Raw8<bf16_t>& r) { r.v = make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ void raw_zero<float>(Raw8<float>& r) {
  r.lo = make_float4(0, 0, 0, 0);
  r.hi = r.lo;
}
template <typename T> __device__ __forceinline__ void raw_load(Raw8<T>& r, const T* p);
template <> __device__ __forceinline__ void raw_load<bf16_t>(Raw8<bf16_t>& r, const bf16_t* p) {
  r.v = *reinterpret_cast<const uint4*>(p);
}
template <> __device__ __forceinline__ void raw_load<float>(Raw8<float>& r, const float* p) {
  r.lo = *reinterpret_cast<const float4*>(p);
  r.hi = *reinterpret_cast<const float4*>(p + 4);
}
template <typename T> __device__ __forceinline__ void raw_unpack(const Raw8<T>& r, float x[8]);
template <> __device__ __forceinline__ void raw_unpack<bf16_t>(const Raw8<bf16_t>& r, float x[8]) {
  x[0] = __uint_as_float(r.v.x << 16); x[1] = __uint_as_float(r.v.x & 0xffff0000u);
  x[2] = __uint_as_float(r.v.y << 16); x[3] = __uint_as_float(r.v.y & 0xffff0000u);
  x[4] = __uint_as_float(r.v.z << 16); x[5] = __uint_as_float(r.v.z & 0xffff0000u);
  x[6] = __uint_as_float(r.v.w << 16); x[7] = __uint_as_float(r.v.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void raw_unpack<float>(const Raw8<float>& r, float x[8]) {
  x[0] = r.lo.x; x[1] = r.lo.y; x[2] = r.lo.z; x[3] = r.lo.w;
  x[4] = r.hi.x; x[5] = r.hi.y; x[6] = r.hi.z; x[7] = r.hi.w;
}

template <typename T, int NT> struct GemmCfg {
  static constexpr int BN = NT * 16;
  static constexpr int LDA = BK + 16 / (int)sizeof(T);
  static constexpr int LDC = BN + 4;
  static constexpr int HALF = BM / 2;
  static constexpr int AB_BYTES = (BM + BN) * LDA * (int)sizeof(T);
  static constexpr int C_BYTES = HALF * LDC * 4;
  static constexpr int TILE_BYTES = ((AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES) + 15) / 16 * 16;
};

// ---- MFMA step over one BK slab held in LDS --------------------------------------------------
template <int NT>
__device__ __forceinline__ void mma_slab(const bf16_t* As, const bf16_t* Bs, int lda, int wave,
                                         int lane, f32x4 (&acc)[2][NT]) {
  const int r16 = lane & 15, kq = (lane >> 4) * 8;
  bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[(wave * 32 + r16) * lda + kq]);
  bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[(wave * 32 + 16 + r16) * lda + kq]);
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[(ni * 16 + r16) * lda + kq]);
    acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[0][ni], 0, 0, 0);
    acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[1][ni], 0, 0, 0);
  }
}
template <int NT>
__device__ __forceinline__ void mma_slab(const float* As, const float* Bs, int lda, int wave,
                                         int lane, f32x4 (&acc)[2][NT]) {
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < BK / 4; ++kk) {
    float a0 = As[(wave * 32 + r16) * lda + kk * 4 + kq];
    float a1 = As[(wave * 32 + 16 + r16) * lda + kk * 4 + kq];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      float b = Bs[(ni * 16 + r16) * lda + kk * 4 + kq];
      acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][ni], 0, 0, 0);
      acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][ni], 0, 0, 0);
    }
  }
}

// ---- rows x small-matrix GEMM: forward (BWD=false) and data gradient (BWD=true) --------------
template <typename T, int NT, bool BWD>
__global__ __launch_bounds__(THREADS) void k_gemm(const GemmArgs a) {
  using C = GemmCfg<T, NT>;
  constexpr int BN = C::BN, LDA = C::LDA, LDC = C::LDC, HALF = C::HALF;
  constexpr int CH = BN / 8;                   // 8-wide chunks per output row
  constexpr int A_TASKS = BM * (BK / 8) / THREADS;  // = 2
  constexpr int B_TASKS = (BN * (BK / 8) + THREADS - 1) / THREADS;

  extern __shared__ __align__(16) unsigned char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + BM * LDA;
  float* Cs = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + C::TILE_BYTES);  // [2][Jp] stat sums

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool want_stats = a.stat_partials != nullptr;
  const bool want_gate = BWD && a.epi.dgate != nullptr;

  for (int i = tid; i < 2 * a.Jp; i += THREADS) red[i] = 0.f;
  __syncthreads();

  const int ntn = (a.J + BN - 1) / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int mt0 = blockIdx.x * a.tiles_per_wg;
  const int mt1 = min(ntm, mt0 + a.tiles_per_wg);

  const T* Bm = reinterpret_cast<const T*>(a.Bm);

  for (int mt = mt0; mt < mt1; ++mt) {
    // per-thread A staging geometry (fixed for the whole m-tile)
    int a_row[A_TASKS], a_kc[A_TASKS], a_img[A_TASKS];
    int64_t a_m[A_TASKS];
#pragma unroll
    for (int i = 0; i < A_TASKS; ++i) {
      const int q = tid + i * THREADS;
      a_row[i] = q >> 2;
      a_kc[i] = (q & 3) * 8;
      a_m[i] = (int64_t)mt * BM + a_row[i];
      a_img[i] = (!BWD && a.tv.gate) ? (int)(a_m[i] / a.hw) : 0;
    }

    for (int nt = 0; nt < ntn; ++nt) {
      f32x4 acc[2][NT];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

      // ---- software-pipelined K loop: raw loads for step k0+BK are issued before the MFMAs of k0
      Raw8<T> ra[A_TASKS], ry[A_TASKS];
      auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_TASKS; ++i) {
          const int k = k0 + a_kc[i];
          raw_zero<T>(ra[i]);
          if (BWD) raw_zero<T>(ry[i]);
          if (a_m[i] < a.M && k < a.R) {
            if (!BWD) {
              raw_load<T>(ra[i], reinterpret_cast<const T*>(a.tv.data) + a_m[i] * a.tv.ld + k);
            } else {
              raw_load<T>(ra[i], reinterpret_cast<const T*>(a.gv.dz) + a_m[i] * a.gv.ld + k);
              if (a.gv.a) raw_load<T>(ry[i], reinterpret_cast<const T*>(a.gv.y) + a_m[i] * a.gv.ld + k);
            }
          }
        }
      };
      auto commit = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_TASKS; ++i) {
          const int k = k0 + a_kc[i];
          float x[8];
          raw_unpack<T>(ra[i], x);
          const bool valid = a_m[i] < a.M && k < a.R;
          if (valid) {
            if (!BWD) {
              ViewCoef vc;
              view_load_coef(a.tv, k, vc);
              view_apply(a.tv, vc, k, a_img[i], x);
            } else if (a.gv.a) {
              float y[8];
              raw_unpack<T>(ry[i], y);
              GradCoef gc;
              grad_load_coef(a.gv, k, gc);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaf(gc.a[e], x[e], fmaf(gc.b[e], y[e], gc.cc[e]));
            }
            if (k + 8 > a.R) {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (k + e >= a.R) x[e] = 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
          }
          store8<T>(&As[a_row[i] * LDA + a_kc[i]], x);
        }
        // weights: raw copy from the L2-resident compute copy
#pragma unroll
        for (int i = 0; i < B_TASKS; ++i) {
          const int q = tid + i * THREADS;
          if (q < BN * (BK / 8)) {
            const int j = q >> 2, kc = (q & 3) * 8;
            const int jj = nt * BN + j, k = k0 + kc;
            float w[8];
            if (jj < a.J && k < a.R) {
              load8<T>(Bm + (size_t)jj * a.ldb + k, w);
              if (k + 8 > a.R) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (k + e >= a.R) w[e] = 0.f;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) w[e] = 0.f;
            }
            store8<T>(&Bs[j * LDA + kc], w);
          }
        }
      };

      issue(0);
      for (int k0 = 0; k0 < a.R; k0 += BK) {
        commit(k0);
        __syncthreads();
        if (k0 + BK < a.R) issue(k0 + BK);
        mma_slab<NT>(As, Bs, LDA, wave, lane, acc);
        __syncthreads();
      }

      // ---- epilogue, two half tiles through LDS so that global I/O is row-major 16 B / lane
      const int cc = (tid % CH) * 8;
      const int j = nt * BN + cc;
      const bool col_ok = j < a.J;
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
      float bias8[8], sc8[8], sh8[8], mean8[8], rstd8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bias8[e] = 0.f; sc8[e] = 1.f; sh8[e] = 0.f; mean8[e] = 0.f; rstd8[e] = 1.f;
      }
      if (col_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (j + e < a.J) {
            if (!BWD && a.bias) bias8[e] = a.bias[j + e];
            if (BWD && a.tv.scale) { sc8[e] = a.tv.scale[j + e]; sh8[e] = a.tv.shift[j + e]; }
            if (BWD && want_stats) { mean8[e] = a.epi.mean[j + e]; rstd8[e] = a.epi.rstd[j + e]; }
          }
        }
      }
      for (int half = 0; half < 2; ++half) {
        if ((wave >> 1) == half) {
          const int wr = (wave & 1) * 32;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                Cs[(wr + mi * 16 + (lane >> 4) * 4 + r) * LDC + ni * 16 + (lane & 15)] = acc[mi][ni][r];
        }
        __syncthreads();
        if (col_ok) {
          for (int q = tid; q < HALF * CH; q += THREADS) {  // THREADS % CH == 0: cc fixed per thread
            const int row = q / CH;
            const int64_t m = (int64_t)mt * BM + half * HALF + row;
            if (m >= a.M) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Cs[row * LDC + cc + e];
            if (!BWD) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (j + e < a.J) ? v[e] + bias8[e] : 0.f;
              store8<T>(reinterpret_cast<T*>(a.out) + m * a.ldo + j, v);
              if (want_stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
              }
            } else {
              const size_t off = (size_t)m * a.tv.ld + j;
              float x[8];
              const bool need_x = (a.tv.act != EDET_ACT_NONE && !want_gate) || want_stats;
              if (need_x) load8<T>(reinterpret_cast<const T*>(a.tv.data) + off, x);
              float g[8];
              if (want_gate) {
                // SE-gated input: the gradient of the gated VALUE is stored as it is (edet_se_gate_bwd takes it from
                // there); the gate's own gradient sums come from k_gate_sums after this kernel, in a fixed order
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = v[e];
              } else if (a.tv.act != EDET_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = v[e] * act_grad_(a.tv.act, fmaf(x[e], sc8[e], sh8[e]));
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = v[e];
              }
              if (a.epi.beta) {
                float old[8];
                load8<T>(reinterpret_cast<const T*>(a.epi.gout) + off, old);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += old[e];
              }
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (j + e >= a.J) g[e] = 0.f;
              store8<T>(reinterpret_cast<T*>(a.epi.gout) + off, g);
              if (want_stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  s1[e] += g[e];
                  s2[e] += g[e] * (x[e] - mean8[e]) * rstd8[e];
                }
              }
            }
          }
        }
        __syncthreads();
      }
      if (want_stats) {
        // the THREADS / CH threads of a column chunk: xor butterfly inside the wave, then the waves one after the other
        // (fixed order, no LDS atomics: the same partial row on every run)
        wave_group_sum(s1, CH);
        wave_group_sum(s2, CH);
        for (int wv = 0; wv < THREADS / 64; ++wv) {
          if (wave == wv && lane < CH && col_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (j + e < a.J) {
                red[j + e] += s1[e];
                red[a.Jp + j + e] += s2[e];
              }
            }
          }
          __syncthreads();
        }
      }
    }  // nt
  }    // mt

  __syncthreads();
  if (want_stats) {
    float* dst = a.stat_partials + (size_t)blockIdx.x * 2 * a.J;
    for (int i = tid; i < 2 * a.J; i += THREADS) {
      const int which = i / a.J, col = i - which * a.J;
      dst[i] = red[which * a.Jp + col];
    }
  }
}

// dgate[img][col] += sum over the pixels of the image of d(view)[m][col] * act(bn(x[m][col])) for an SE-gated input
// view, from the gradient k_gemm has just stored.  Workgroup = (image, 64 columns); thread = (8-column chunk, one of 32
// pixel slices); the slices are combined by wave_group_sum and then in wave order -- one writer per element, a fixed
// order: the same bits on every run (the in-kernel version added its sums with LDS / global atomics).
template <typename T>
__global__ __launch_bounds__(THREADS) void k_gate_sums(const GemmArgs a) {
  __shared__ float part[THREADS / 64][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunk = (a.J + 63) / 64;
  const int img = blockIdx.x / nchunk, j = (blockIdx.x % nchunk) * 64 + (tid & 7) * 8;
  const int slice = tid >> 3;
  float gp[8], sc8[8], sh8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gp[e] = 0.f; sc8[e] = 1.f; sh8[e] = 0.f; }
  if (j < a.J) {
    if (a.tv.scale) { loadf8(a.tv.scale + j, sc8); loadf8(a.tv.shift + j, sh8); }     // (Jp-padded rows: c % 8 == 0)
    for (int r = slice; r < a.hw; r += THREADS / 8) {
      const size_t off = ((size_t)img * a.hw + r) * a.tv.ld + j;
      float x[8], v[8];
      load8<T>(reinterpret_cast<const T*>(a.tv.data) + off, x);
      load8<T>(reinterpret_cast<const T*>(a.epi.gout) + off, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) gp[e] = fmaf(v[e], act_apply_(a.tv.act, fmaf(x[e], sc8[e], sh8[e])), gp[e]);
    }
  }
  wave_group_sum(gp, 8);
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][lane * 8 + e] = gp[e];
  }
  __syncthreads();
  if (tid < 64) {
    const int col = (blockIdx.x % nchunk) * 64 + tid;
    if (col < a.J) {
      float t = part[0][tid];
      for (int w = 1; w < THREADS / 64; ++w) t += part[w][tid];
      a.epi.dgate[(size_t)img * a.J + col] += t;
    }
  }
}

template <typename T, bool BWD>
int launch_gemm(GemmArgs& a, int* nparts_out, hipStream_t st) {
  const int ntm = cdiv(a.M, BM);
  a.tiles_per_wg = cdiv(ntm, EDET_MAX_PARTS);
  const int grid = cdiv(ntm, a.tiles_per_wg);
  a.Jp = (a.J + 7) / 8 * 8;
  const bool gate = BWD && a.epi.dgate != nullptr;
  EDET_CHECK(!(gate && a.epi.beta), "edet_pw_bwd_data: the gate gradient of an SE-gated input needs beta == 0 (its sums are "
             "taken from the stored gradient)");
  const size_t extra = (size_t)2 * a.Jp * sizeof(float);
  if (nparts_out) *nparts_out = grid;
  if (a.J <= 32) {
    edet_launch(k_gemm<T, 2, BWD>, dim3(grid), dim3(THREADS), GemmCfg<T, 2>::TILE_BYTES + extra, st, a);
  } else if (a.J <= 64) {
    edet_launch(k_gemm<T, 4, BWD>, dim3(grid), dim3(THREADS), GemmCfg<T, 4>::TILE_BYTES + extra, st, a);
  } else {
    edet_launch(k_gemm<T, 8, BWD>, dim3(grid), dim3(THREADS), GemmCfg<T, 8>::TILE_BYTES + extra, st, a);
  }
  EDET_LAUNCH_CHECK(BWD ? "edet_pw_bwd_data" : "edet_pw_fwd");
  if (gate) {
    edet_launch(k_gate_sums<T>, dim3(a.tv.n * cdiv(a.J, 64)), dim3(THREADS), 0, st, a);
    EDET_LAUNCH_CHECK("edet_pw_bwd_data (gate sums)");
  }
  return 0;
}

// ---- weight gradient: P[i][j] = sum_m U[m][i] * V[m][j] -----------------------------------------
// U is the operand with more channels (tiled 64 columns per workgroup, one 16-col MFMA tile per
// wave), V the one with fewer (all NJ*16 columns held per workgroup).  Either may be the
// activated input view or the gradient view; the big tensor is therefore read exactly once.
struct WgradArgs {
  edet_tview_t tv;  // conv input (activated view), K = tv.c channels
  edet_gview_t gv;  // dy, N = gv.c channels
  int u_is_grad;    // 1: U = dy (i -> n, j -> k); 0: U = in (i -> k, j -> n)
  int M, CU, CV;    // rows, channels of U and V
  int hw;
  int rows_per_wg;
  float* dw;        // [K][N] fp32
  int N;            // = gv.c (row length of dw)
  float* ws;        // row splits > 1: partial sums [split][K][N], added in split order by edet_reduce_partials
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int BMR = 64; };
template <> struct WgCfg<float> { static constexpr int BMR = 32; };

template <typename T>
__device__ __forceinline__ void wg_load(const WgradArgs& a, bool is_grad, int64_t m, int c, float x[8]) {
  if (is_grad) {
    GradCoef gc;
    grad_load_coef(a.gv, c, gc);
    grad_load<T>(a.gv, gc, (size_t)m * a.gv.ld + c, x);
  } else {
    load8<T>(reinterpret_cast<const T*>(a.tv.data) + (size_t)m * a.tv.ld + c, x);
    ViewCoef vc;
    view_load_coef(a.tv, c, vc);
    view_apply(a.tv, vc, c, a.tv.gate ? (int)(m / a.hw) : 0, x);
  }
}

template <typename T, int NJ>
__global__ __launch_bounds__(THREADS) void k_wgrad(const WgradArgs a) {
  constexpr int BMR = WgCfg<T>::BMR;
  constexpr int TI = 64, TJ = NJ * 16;
  constexpr bool IS_BF = sizeof(T) == 2;
  // bf16: transposed images Ut[TI][BMR+8], Vt[TJ][BMR+8]; fp32: row-major Us[BMR][TI+4], Vs[BMR][TJ+4]
  constexpr int LDT = BMR + 8;
  constexpr int LDU = TI + 4, LDV = TJ + 4;
  extern __shared__ __align__(16) unsigned char smem[];
  T* Ub = reinterpret_cast<T*>(smem);
  T* Vb = IS_BF ? Ub + TI * LDT : Ub + BMR * LDU;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nti = (a.CU + TI - 1) / TI;
  const int it = blockIdx.x % nti;
  const int split = blockIdx.x / nti;
  const int i0 = it * TI;
  const int64_t r0 = (int64_t)split * a.rows_per_wg;
  const int64_t r1 = min((int64_t)a.M, r0 + a.rows_per_wg);
  const bool u_grad = a.u_is_grad != 0;

  f32x4 acc[NJ];
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) acc[nj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int64_t mb = r0; mb < r1; mb += BMR) {
    if constexpr (IS_BF) {
      // task = 8 rows x 8 channels, transposed in registers while packing to bf16
      constexpr int UT = (BMR / 8) * (TI / 8), VT = (BMR / 8) * (TJ / 8);
      for (int q = tid; q < UT + VT; q += THREADS) {
        const bool isU = q < UT;
        const int qq = isU ? q : q - UT;
        const int ncol = isU ? TI / 8 : TJ / 8;
        const int cb = (qq % ncol) * 8;      // channel offset within the tile
        const int rb = (qq / ncol) * 8;      // row offset within the slab
        const int cglob = (isU ? i0 : 0) + cb;
        const int cmax = isU ? a.CU : a.CV;
        float f[8][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int64_t m = mb + rb + r;
          if (m < r1 && cglob < cmax) {
            wg_load<T>(a, isU ? u_grad : !u_grad, m, cglob, f[r]);
            if (cglob + 8 > cmax) {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (cglob + e >= cmax) f[r][e] = 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[r][e] = 0.f;
          }
        }
        bf16_t* dst = reinterpret_cast<bf16_t*>(isU ? Ub : Vb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          uint4 o;
          o.x = pack2bf(f[0][e], f[1][e]);
          o.y = pack2bf(f[2][e], f[3][e]);
          o.z = pack2bf(f[4][e], f[5][e]);
          o.w = pack2bf(f[6][e], f[7][e]);
          *reinterpret_cast<uint4*>(&dst[(cb + e) * LDT + rb]) = o;
        }
      }
      __syncthreads();
      const bf16_t* Ut = reinterpret_cast<const bf16_t*>(Ub);
      const bf16_t* Vt = reinterpret_cast<const bf16_t*>(Vb);
#pragma unroll
      for (int ks = 0; ks < BMR / 32; ++ks) {
        const int kq = ks * 32 + (lane >> 4) * 8;
        bf16x8 af = *reinterpret_cast<const bf16x8*>(&Ut[(wave * 16 + (lane & 15)) * LDT + kq]);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Vt[(nj * 16 + (lane & 15)) * LDT + kq]);
          acc[nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[nj], 0, 0, 0);
        }
      }
      __syncthreads();
    } else {
      constexpr int UT = BMR * (TI / 8), VT = BMR * (TJ / 8);
      for (int q = tid; q < UT + VT; q += THREADS) {
        const bool isU = q < UT;
        const int qq = isU ? q : q - UT;
        const int ncol = isU ? TI / 8 : TJ / 8;
        const int cb = (qq % ncol) * 8;
        const int row = qq / ncol;
        const int cglob = (isU ? i0 : 0) + cb;
        const int cmax = isU ? a.CU : a.CV;
        const int64_t m = mb + row;
        float f[8];
        if (m < r1 && cglob < cmax) {
          wg_load<T>(a, isU ? u_grad : !u_grad, m, cglob, f);
          if (cglob + 8 > cmax) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (cglob + e >= cmax) f[e] = 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        float* dst = reinterpret_cast<float*>(isU ? Ub : Vb);
        store8<float>(&dst[row * (isU ? LDU : LDV) + cb], f);
      }
      __syncthreads();
      const float* Us = reinterpret_cast<const float*>(Ub);
      const float* Vs = reinterpret_cast<const float*>(Vb);
#pragma unroll
      for (int ks = 0; ks < BMR / 4; ++ks) {
        const int mrow = ks * 4 + (lane >> 4);
        const float af = Us[mrow * LDU + wave * 16 + (lane & 15)];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          const float bfr = Vs[mrow * LDV + nj * 16 + (lane & 15)];
          acc[nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfr, acc[nj], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // D layout: row i = (lane>>4)*4 + r, col j = lane & 15
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 16 + (lane >> 4) * 4 + r;
      const int j = nj * 16 + (lane & 15);
      if (i < a.CU && j < a.CV) {
        const int k = u_grad ? j : i;
        const int n = u_grad ? i : j;
        // one writer per element: this split's partial row, or (a single split) dW itself -- no atomics
        if (a.ws) a.ws[((size_t)split * a.tv.c + k) * a.N + n] = acc[nj][r];
        else a.dw[(size_t)k * a.N + n] += acc[nj][r];
      }
    }
  }
}

template <typename T, int NJ>
void launch_wgrad_nj(const WgradArgs& a, int grid, hipStream_t st) {
  constexpr int BMR = WgCfg<T>::BMR;
  constexpr int TJ = NJ * 16;
  const size_t lds = sizeof(T) == 2 ? (size_t)(64 + TJ) * (BMR + 8) * 2
                                     : (size_t)BMR * ((64 + 4) + (TJ + 4)) * 4;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, NJ>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
  }
  edet_launch(k_wgrad<T, NJ>, dim3(grid), dim3(THREADS), lds, st, a);
}

template <typename T>
int launch_wgrad(WgradArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int K = a.tv.c, N = a.gv.c;
  a.N = N;
  a.u_is_grad = N >= K ? 1 : 0;
  a.CU = a.u_is_grad ? N : K;
  a.CV = a.u_is_grad ? K : N;
  EDET_CHECK(a.CV <= 640, "edet_pw_bwd_weight: min(cin, cout) = %d > 640 unsupported", a.CV);
  const int nti = cdiv(a.CU, 64);
  // ~1024 workgroups; each at least 512 rows
  int split = 1024 / nti;
  // the row splits hand their partial sums over through the workspace (ordered reduction, the same bits on every run);
  // without one a single split adds into dW directly
  const size_t row_bytes = (size_t)K * N * sizeof(float);
  const int ws_rows = workspace ? (int)(workspace_bytes / row_bytes < 4096 ? workspace_bytes / row_bytes : 4096) : 0;
  if (split > ws_rows) split = ws_rows;
  if (split < 1) split = 1;
  int64_t rows = cdiv(a.M, split);
  if (rows < 512) rows = 512;
  rows = (rows + 63) / 64 * 64;
  a.rows_per_wg = (int)rows;
  split = cdiv(a.M, rows);
  a.ws = split > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  const int grid = nti * split;
  if (a.CV <= 64) launch_wgrad_nj<T, 4>(a, grid, st);
  else if (a.CV <= 128) launch_wgrad_nj<T, 8>(a, grid, st);
  else if (a.CV <= 192) launch_wgrad_nj<T, 12>(a, grid, st);
  else if (a.CV <= 320) launch_wgrad_nj<T, 20>(a, grid, st);
  else launch_wgrad_nj<T, 40>(a, grid, st);
  EDET_LAUNCH_CHECK("edet_pw_bwd_weight");
  if (a.ws && edet_reduce_partials(a.ws, split, (int64_t)K * N, a.dw, st) != 0) return -2;
  return 0;
}

}  // namespace

// workgroup-tiled bf16 kernels for the wide layers (pw_big.hip); same return convention
int pwb_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, hipStream_t st);
int pwb_try_dgrad(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st);
int pwb_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st);

// Which bf16 implementation serves a (rows, cin, cout) pointwise layer.  The streaming kernels own the
// HBM-bound layers (few channels, many rows); the tiled kernels own the layers whose weight matrix is large
// (thresholds below).  EDET_PW_IMPL = stream | big | tiled forces one implementation where its envelope allows
// (every value is exercised by the parity tests); it is read per call.
enum { PW_AUTO = 0, PW_STREAM = 1, PW_BIG = 2, PW_TILED = 3 };
static int pw_impl_env() {
  const char* e = getenv("EDET_PW_IMPL");
  if (!e || !*e) return PW_AUTO;
  if (!strcmp(e, "stream")) return PW_STREAM;
  if (!strcmp(e, "big")) return PW_BIG;
  if (!strcmp(e, "tiled")) return PW_TILED;
  return PW_AUTO;
}
// Which implementation goes first: the workgroup-tiled kernels (pw_big.hip) once cin*cout reaches a threshold,
// else the wave-private streaming kernels.  Thresholds from A/B runs of the D0 640x640 step (r01g, per-op
// totals in ms at threshold 2048 / 4096 / 8192 / 24576: forward 8.54 / 7.98 / 7.87 / 8.05, data gradient
// 12.15 / 12.85 / 12.88 / 13.4, weight gradient 14.21 / 14.26 / 14.51 / 14.4).
enum { PW_OP_FWD = 0, PW_OP_DGRAD = 1, PW_OP_WGRAD = 2 };
// r03e (lab, per layer, EDET_PW_IMPL=big against stream with the round-3 workgroup targets): the weight gradient of
// the 96 -> 24 and 144 -> 24 project layers on the 160-row maps is 7 / 22 % faster tiled (0.42 / 0.80 ms against
// 0.45 / 1.02 ms) -> threshold 4096 -> 2048; the forward 64 -> 64 layers of the 20x20 and smaller levels take 9 us
// tiled against 19 us streamed -> tiled from 4096 when the map has at most 64 K rows.
static bool pw_prefers_big(int op, int64_t rows, int cin, int cout) {
  static const int64_t minkn[3] = {8192, 2048, 2048};
  const int64_t kn = (int64_t)cin * cout;
  if (op == PW_OP_FWD && kn >= 4096 && rows <= 65536 && rows >= 1024) return true;
  return kn >= minkn[op] && rows >= 1024;
}

// streaming bf16 kernels (pw_stream.hip); return 1 = handled, 0 = shape outside their envelope
int pws_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, hipStream_t st);

int pws_try_dgrad(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                  const edet_bwd_epi_t* epi, int* nparts_out, hipStream_t st);

extern "C" int edet_pw_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias,
                           void* out, int cout, int ldo, float* stat_partials, int* nparts_out,
                           int dtype, void* stream) {
  EDET_CHECK(in && in->data && wt && out, "edet_pw_fwd: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && ldo % 8 == 0 && ldw % 8 == 0 && ldw >= in->c,
             "edet_pw_fwd: channel counts/strides must be multiples of 8 (c=%d ld=%d ldo=%d ldw=%d)",
             in->c, in->ld, ldo, ldw);
  EDET_CHECK(ldo >= cout && cout <= 1152 * 4, "edet_pw_fwd: bad cout/ldo");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = wt; a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = in->c; a.J = cout; a.hw = in->h * in->w;
  a.bias = bias; a.out = out; a.ldo = ldo; a.stat_partials = stat_partials;
  if (dtype == EDET_BF16) {
    const int impl = pw_impl_env();
    const bool big_first = impl == PW_BIG || (impl == PW_AUTO && pw_prefers_big(PW_OP_FWD, a.M, in->c, cout));
    int rc = 0;
    if (big_first) rc = pwb_try_fwd(in, wt, ldw, bias, out, cout, ldo, stat_partials, nparts_out, to_stream(stream));
    if (rc == 0 && impl != PW_TILED && impl != PW_BIG)
      rc = pws_try_fwd(in, wt, ldw, bias, out, cout, ldo, stat_partials, nparts_out, to_stream(stream));
    if (rc == 0 && !big_first && impl == PW_AUTO && (int64_t)in->c * cout >= 4096)
      rc = pwb_try_fwd(in, wt, ldw, bias, out, cout, ldo, stat_partials, nparts_out, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
    return launch_gemm<bf16_t, false>(a, nparts_out, to_stream(stream));
  }
  if (dtype == EDET_F32) return launch_gemm<float, false>(a, nparts_out, to_stream(stream));
  EDET_CHECK(false, "edet_pw_fwd: bad dtype %d", dtype);
}

int pwb_fwd_f32out(const edet_tview_t* in, const void* wt, int ldw, const float* bias, float* out, int cout, int ldo,
                   hipStream_t st);

extern "C" int edet_pw_fwd_f32out(const edet_tview_t* in, const void* wt, int ldw, const float* bias, float* out,
                                  int cout, int ldo, void* stream) {
  EDET_CHECK(in && in->data && wt && out, "edet_pw_fwd_f32out: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && ldw % 8 == 0 && ldw >= in->c && ldo % 8 == 0 && ldo >= cout,
             "edet_pw_fwd_f32out: channel counts / strides must be multiples of 8 (c=%d ld=%d ldo=%d ldw=%d)", in->c, in->ld,
             ldo, ldw);
  const int rc = pwb_fwd_f32out(in, wt, ldw, bias, out, cout, ldo, to_stream(stream));
  EDET_CHECK(rc != 0, "edet_pw_fwd_f32out: shape outside the tiled kernel's envelope");
  return rc < 0 ? rc : 0;
}

extern "C" int edet_pw_bwd_data(const edet_gview_t* dy, const void* w, int ldw,
                                const edet_tview_t* in, const edet_bwd_epi_t* epi, int* nparts_out,
                                int dtype, void* stream) {
  EDET_CHECK(dy && dy->dz && w && in && in->data && epi && epi->gout, "edet_pw_bwd_data: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && dy->ld % 8 == 0 && ldw % 8 == 0 && ldw >= dy->c,
             "edet_pw_bwd_data: strides must be multiples of 8");
  EDET_CHECK(!(epi->stat_partials && epi->beta), "edet_pw_bwd_data: fused stats need beta == 0");
  EDET_CHECK(!(epi->dgate && !in->gate), "edet_pw_bwd_data: dgate given but input view has no gate");
  // (checked here, in front of the dispatch: the same call must not pass or fail by the implementation it lands on)
  EDET_CHECK(!(in->gate && epi->dgate && epi->beta), "edet_pw_bwd_data: the gate gradient of an SE-gated input needs beta == 0");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy;
  a.Bm = w; a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = dy->c; a.J = in->c; a.hw = in->h * in->w;
  a.epi = *epi; a.stat_partials = epi->stat_partials;
  if (dtype == EDET_BF16) {
    const int impl = pw_impl_env();
    const bool big_first = impl == PW_BIG || (impl == PW_AUTO && pw_prefers_big(PW_OP_DGRAD, a.M, in->c, dy->c));
    int rc = 0;
    // SE-gated input: the tuned kernels store the gradient of the gated value and leave the gate-gradient sums to
    // k_gate_sums below (one writer per element; r06: their own sums were floating-point atomics, the last of the bf16
    // training step) -- the generic kernel (launch_gemm) does the same on its own
    edet_bwd_epi_t e2 = *epi;
    if (epi->dgate) e2.flags |= EDET_EPI_GATE_SUMS_LATER;
    if (big_first) rc = pwb_try_dgrad(dy, w, ldw, in, &e2, nparts_out, to_stream(stream));
    if (rc == 0 && impl != PW_TILED && impl != PW_BIG)
      rc = pws_try_dgrad(dy, w, ldw, in, &e2, nparts_out, to_stream(stream));
    if (rc == 0 && !big_first && impl == PW_AUTO && (int64_t)in->c * dy->c >= 4096)
      rc = pwb_try_dgrad(dy, w, ldw, in, &e2, nparts_out, to_stream(stream));
    if (rc > 0 && epi->dgate) {
      edet_launch(k_gate_sums<bf16_t>, dim3(a.tv.n * cdiv(a.J, 64)), dim3(THREADS), 0, to_stream(stream), a);
      EDET_LAUNCH_CHECK("edet_pw_bwd_data (gate sums)");
    }
    if (rc != 0) return rc < 0 ? rc : 0;
    return launch_gemm<bf16_t, true>(a, nparts_out, to_stream(stream));
  }
  if (dtype == EDET_F32) return launch_gemm<float, true>(a, nparts_out, to_stream(stream));
  EDET_CHECK(false, "edet_pw_bwd_data: bad dtype %d", dtype);
}

int pws_try_wgrad(const edet_tview_t* in, const edet_gview_t* dy, float* dweight, void* workspace,
                  size_t workspace_bytes, hipStream_t st);

extern "C" int edet_pw_bwd_weight(const edet_tview_t* in, const edet_gview_t* dy, float* dweight,
                                  void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  EDET_CHECK(in && in->data && dy && dy->dz && dweight, "edet_pw_bwd_weight: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && dy->ld % 8 == 0, "edet_pw_bwd_weight: strides % 8");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.tv = *in; a.gv = *dy;
  a.M = in->n * in->h * in->w; a.hw = in->h * in->w;
  a.dw = dweight;
  if (dtype == EDET_BF16) {
    const int impl = pw_impl_env();
    const bool big_first = impl == PW_BIG || (impl == PW_AUTO && pw_prefers_big(PW_OP_WGRAD, a.M, in->c, dy->c));
    int rc = 0;
    if (big_first) rc = pwb_try_wgrad(in, dy, dweight, workspace, workspace_bytes, to_stream(stream));
    if (rc == 0 && impl != PW_TILED && impl != PW_BIG)
      rc = pws_try_wgrad(in, dy, dweight, workspace, workspace_bytes, to_stream(stream));
    if (rc == 0 && !big_first && impl == PW_AUTO && (int64_t)in->c * dy->c >= 4096)
      rc = pwb_try_wgrad(in, dy, dweight, workspace, workspace_bytes, to_stream(stream));
    if (rc != 0) return rc < 0 ? rc : 0;
    return launch_wgrad<bf16_t>(a, workspace, workspace_bytes, to_stream(stream));
  }
  if (dtype == EDET_F32) return launch_wgrad<float>(a, workspace, workspace_bytes, to_stream(stream));
  EDET_CHECK(false, "edet_pw_bwd_weight: bad dtype %d", dtype);
}

// streaming fused data + weight gradient (pw_stream.hip); return 1 = handled, 0 = shape outside its envelope
int pws_try_bwd_fused(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                      const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                      size_t workspace_bytes, hipStream_t st);

// one-pass tiled data + weight gradient (pw_tile_bwd.hip); same return convention
int pwt_try_bwd(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in, const edet_bwd_epi_t* epi,
                int* nparts_out, float* dweight, void* workspace, size_t workspace_bytes, hipStream_t st);

extern "C" int edet_pw_bwd(const edet_gview_t* dy, const void* w, int ldw, const edet_tview_t* in,
                           const edet_bwd_epi_t* epi, int* nparts_out, float* dweight, void* workspace,
                           size_t workspace_bytes, int dtype, void* stream) {
  EDET_CHECK(dy && dy->dz && w && in && in->data && epi && epi->gout && dweight, "edet_pw_bwd: null pointer");
  EDET_CHECK(in->c % 8 == 0 && in->ld % 8 == 0 && dy->ld % 8 == 0 && ldw % 8 == 0 && ldw >= dy->c,
             "edet_pw_bwd: strides must be multiples of 8");
  EDET_CHECK(!(epi->stat_partials && epi->beta), "edet_pw_bwd: fused stats need beta == 0");
  EDET_CHECK(!(epi->dgate && !in->gate), "edet_pw_bwd: dgate given but input view has no gate");
  EDET_CHECK(!(in->gate && epi->dgate && epi->beta), "edet_pw_bwd: the gate gradient of an SE-gated input needs beta == 0");
  if (dtype == EDET_BF16) {
    const int impl = pw_impl_env();
    if (impl == PW_AUTO || impl == PW_STREAM) {
      const int rc = pws_try_bwd_fused(dy, w, ldw, in, epi, nparts_out, dweight, workspace, workspace_bytes,
                                       to_stream(stream));
      if (rc != 0) return rc < 0 ? rc : 0;
    }
    // the one-pass tiled kernel (pw_tile_bwd.hip): cout <= 128; EDET_PWT=0 switches it off (lab / test switch, read per call)
    const char* pwt_env = getenv("EDET_PWT");
    if (impl == PW_AUTO && !(pwt_env && pwt_env[0] == '0')) {
      const int rc = pwt_try_bwd(dy, w, ldw, in, epi, nparts_out, dweight, workspace, workspace_bytes, to_stream(stream));
      if (rc != 0) return rc < 0 ? rc : 0;
    }
  }
  const int rc = edet_pw_bwd_weight(in, dy, dweight, workspace, workspace_bytes, dtype, stream);
  if (rc != 0) return rc;
  return edet_pw_bwd_data(dy, w, ldw, in, epi, nparts_out, dtype, stream);
}
