// Wide pointwise (1x1) convolution, forward, with both operands PREFETCHED THROUGH LDS by the gfx950 LDS-DMA.
//
// k_big_gemm (pw_big.hip) stages its operands through registers: one register set of raw loads is in flight across ONE
// MFMA phase, the kernel sits at 241 VGPRs at two workgroups per CU, and any deeper register prefetch spills (r06aa).
// This kernel takes the registers out of the load path:
//
//   * global_load_lds_dwordx4 (1 KiB per wave instruction, lane-linear destination) brings the RAW 128 x BK chunk of the
//     streamed operand, the 128 x BK weight chunk and a block of per-channel coefficients (BatchNorm scale / shift, the
//     SE gate rows of the up to four images of the tile; one copy per wave) into one of NST = 4 LDS stages: the loads of THREE
//     reduction steps are in flight, counted by hand (s_waitcnt vmcnt(2 G) before a stage is touched);
//   * chunk c of row r lives in slot c ^ swz(r) (the permutation is applied to the per-lane GLOBAL address, the LDS image
//     of an LDS-DMA is fixed): ds_read_b128 fragment reads without padding;
//   * the producer's BatchNorm / swish / SE gate is applied IN PLACE in LDS, by the thread that requested the chunk (its
//     own wave's DMA, its own wave's coefficient copy: no cross-wave dependency before the step's single barrier), one
//     step ahead of the MFMAs;
//   * raw s_barrier + lgkmcnt(0) (__syncthreads() would drain the DMA queue), and every LDS access of the loop through a
//     __restrict__ parameter (see lds_ld16): the compiler otherwise waits vmcnt(0) before each of them.
//
// Same tile (128 x 128, 2 x 2 waves of 2 x 2 v_mfma_f32_32x32x16_bf16), same operand values, same accumulation order and
// the same epilogue as k_big_gemm<false, false>: the results are BIT-IDENTICAL to that kernel's
// (tests/test_gpu_kernels.py::test_pw_fwd_glds_equals_register_staged).
//
// Measured (r06ak / r06al, LABNOTES.md): with 64-deep steps the four 36 KiB stages leave ONE workgroup per CU and the
// kernel loses 15-60 % to the register-staged one (nothing runs under a workgroup's prologue and epilogue); with 32-deep
// steps (18 KiB stages, two workgroups per CU) it wins 5-19 % on the SE-gated project layers and ties elsewhere.  What
// the A/B against PLAIN views showed is that these layers are bound by the on-load transform itself (two transcendentals
// per element and column tile: a swish + gate view costs 2 x a stored tensor in BOTH kernels), not by load latency --
// so the dispatcher (pwb_try_fwd) sends the gated views here and keeps everything else where it was.
//
// Reference call sites replaced: tf.keras.layers.Conv2D 1x1 in efficientdet/backbone/efficientnet_model.py:345-353
// (project) with the preceding BatchNorm / activation / SE multiply (:183-195, :378-392).
#include <stdlib.h>

#include "common.h"

namespace pwg {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int THREADS = 256;
constexpr int BM = 128, BJ = 128;
constexpr int NST = 4;
constexpr int LDC_BF = BJ * 2 + 16;              // bf16 C tile row stride (272)
// Geometry of a reduction step of BK elements.  BK = 32 (the instantiated one): 18 KiB stages, two workgroups per CU (one's
// prologue / epilogue under the other's main loop).  BK = 64: 36 KiB stages, one workgroup per CU -- measured slower.
template <int BK> struct Geo {
  static constexpr int CPR = BK / 8;                  // 16-byte chunks per row
  static constexpr int RPD = 64 / CPR;                // rows per DMA instruction (1 KiB)
  static constexpr int NI = BM / RPD / 4;             // DMA instructions per wave, operand and step
  static constexpr int ROWB = BK * 2;                 // bytes per row
  static constexpr int RP = 128 / ROWB;               // rows per 128 bytes of LDS
  static constexpr int TILE_BYTES = BM * ROWB;
  static constexpr int NG = BK == 32 ? 4 : 2;         // gate rows (images a row tile may touch): 4 -> maps down to 7 x 7
  static constexpr int NT = 2 + NG;                   // tables of a coefficient block: scale | shift | gate(img0 ..)
  static_assert(NT * (BK / 4) <= 64, "one DMA instruction per coefficient block");
  static constexpr int CW = NT * BK * 4;              // coefficient block of one wave
  static constexpr int STAGE_BYTES = 2 * TILE_BYTES + 4 * CW;
  static constexpr int SMEM_BYTES = NST * STAGE_BYTES;
  static_assert(BM * LDC_BF <= SMEM_BYTES, "the C tile must fit in the stages");
  // chunk c of row r lives in slot c ^ swz(r)
  __device__ static __forceinline__ int swz(int r) { return (r / RP) & (CPR - 1); }
};

struct Args {
  edet_tview_t tv;
  const bf16_t* Bm;   // [J][ldb], reduction index contiguous
  int ldb;
  int M, R, J;        // rows, reduction length, output columns
  int hw;             // pixels per image (>= 43 when gated: a row tile touches at most four images)
  int ntm, ntj, tpw, ngrp;
  const float* bias;
  bf16_t* out;
  int ldo;
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

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
// active lanes x 16 B from per-lane global addresses to LDS dst + 16 lane; dst is wave-uniform
__device__ __forceinline__ void dma16(const void* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
}
// LDS accesses of the main loop go through __restrict__ parameters: after inlining they carry alias-scope metadata, and
// the compiler's wait-count pass (which otherwise puts s_waitcnt vmcnt(0) in front of EVERY LDS access that follows an
// LDS-DMA) then only waits for DMAs it can prove to alias -- none; the waits are the counted ones below.  (Stores that the
// optimiser merges from several inlined calls lose the scope again: one store call site per operand.)
__device__ __attribute__((always_inline)) inline uint4 lds_ld16(const uint4* __restrict__ p) { return *p; }
__device__ __attribute__((always_inline)) inline void lds_st16(uint4* __restrict__ p, uint4 v) { *p = v; }
__device__ __attribute__((always_inline)) inline bf16x8 lds_ldf(const bf16x8* __restrict__ p) { return *p; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
  const uint4 lo = lds_ld16(reinterpret_cast<const uint4*>(p)), hi = lds_ld16(reinterpret_cast<const uint4*>(p + 4));
  v[0] = __uint_as_float(lo.x); v[1] = __uint_as_float(lo.y); v[2] = __uint_as_float(lo.z); v[3] = __uint_as_float(lo.w);
  v[4] = __uint_as_float(hi.x); v[5] = __uint_as_float(hi.y); v[6] = __uint_as_float(hi.z); v[7] = __uint_as_float(hi.w);
}

// COEF: the view carries BatchNorm scale / shift and / or an SE gate (one more DMA per wave and step)
template <int BK, bool COEF, bool SWISH>
__global__ __launch_bounds__(THREADS, BK == 64 ? 1 : 2) void k_wide_fwd(const Args a) {
  using GE = Geo<BK>;
  constexpr int NI = GE::NI, TILE_BYTES = GE::TILE_BYTES, STAGE_BYTES = GE::STAGE_BYTES, CW = GE::CW, ROWB = GE::ROWB;
  extern __shared__ __align__(1024) unsigned char smem[];
  constexpr int G = 2 * NI + (COEF ? 1 : 0);      // DMA instructions per wave and reduction step
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wj = wave >> 1;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int jt = q % a.ntj;
  const int grp = (q / a.ntj) * 8 + xcd;
  if (grp >= a.ngrp) return;
  const int j0 = jt * BJ;
  const bool want_stats = a.stat_partials != nullptr;
  const bool affine = a.tv.scale != nullptr, gated = a.tv.gate != nullptr;
  const int nk = (a.R + BK - 1) / BK;
  const bool ragged = a.R % BK != 0;

  // DMA geometry: instruction i of wave w fills rows (NI w + i) RPD .. + RPD; lane -> row lane / CPR, slot lane % CPR,
  // which holds chunk cch = slot ^ swz(row) (the instruction's first row is a multiple of 8: swz(row) = swz(lrow))
  const int lrow = lane / GE::CPR;
  const int cch = (lane % GE::CPR) ^ GE::swz(lrow);
  // epilogue geometry: thread -> 8 output columns ec*8.., rows er + 16*i
  const int ec = tid & 15, er = tid >> 4;
  const int ej = j0 + ec * 8;
  const bool ecol_ok = ej < a.J;

  const bf16_t* SRC = reinterpret_cast<const bf16_t*>(a.tv.data);
  float tot1 = 0.f, tot2 = 0.f;

  const int mt_end = min(a.ntm, (grp + 1) * a.tpw);
  for (int mt = grp * a.tpw; mt < mt_end; ++mt) {
    const int m0 = mt * BM;
    const bf16_t* arow[NI];
    const bf16_t* brow[NI];
    unsigned gsel = 0, bzero = 0;                          // gsel: 2 bits per chunk = the row's image - img0
    const int img0 = m0 / a.hw;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (wave * NI + i) * GE::RPD + lrow;
      const int m = min(m0 + r, a.M - 1);                 // rows past M re-read row M-1 (never stored)
      arow[i] = SRC + (size_t)m * a.tv.ld;
      if (gated) gsel |= (unsigned)(m / a.hw - img0) << (2 * i);
      const int j = j0 + r;
      if (j >= a.J) bzero |= 1u << i;                     // weight rows past J: zeroed in LDS by their owner
      brow[i] = a.Bm + (size_t)min(j, a.J - 1) * a.ldb;
    }
    // coefficient block of this wave: NT tables x BK floats; lane -> table lane / (BK / 4), 4 floats at lane % (BK / 4)
    // (the lanes past the last table sit the instruction out)
    const float* crow = nullptr;
    const bool clane = lane < GE::NT * (BK / 4);
    if (COEF) {
      const int b = min(lane / (BK / 4), GE::NT - 1);
      // a missing table reads the other one (its values are not used); images past the last one re-read the last
      const float* sc = affine ? a.tv.scale : a.tv.gate;
      const float* sh = affine ? a.tv.shift : a.tv.gate;
      const float* gr = gated ? a.tv.gate + (size_t)min(img0 + max(b - 2, 0), a.tv.n - 1) * a.R : a.tv.scale;
      crow = b == 0 ? sc : (b == 1 ? sh : gr);
    }

    auto issue = [&](int kt, unsigned char* stage) {
      const int k = kt * BK + cch * 8;
      const int kc = k < a.R ? k : 0;                      // chunks past the reduction length: any finite data, zeroed / unused
      unsigned char* As = stage + wave * (NI * 1024);
      unsigned char* Bs = stage + TILE_BYTES + wave * (NI * 1024);
#pragma unroll
      for (int i = 0; i < NI; ++i) dma16(arow[i] + kc, As + i * 1024);
#pragma unroll
      for (int i = 0; i < NI; ++i) dma16(brow[i] + kc, Bs + i * 1024);
      if (COEF) {
        const int kk = min(kt * BK + (lane % (BK / 4)) * 4, a.R - 4);
        if (clane) dma16(crow + kk, stage + 2 * TILE_BYTES + wave * CW);
      }
    };
    // in-place transform of the thread's own chunks of the streamed operand (and zeroing of weight rows past J).
    // Chunks past the reduction length (last step of a length that is not a multiple of BK) become zeros.  ONE instance
    // of every LDS access (no tail / no-tail instantiations: merged accesses lose their alias scope).
    auto transform_step = [&](int kt, unsigned char* stage) {
      const int k = kt * BK + cch * 8;
      const bool kok = k < a.R;
      unsigned char* Ap = stage + wave * (NI * 1024) + lane * 16;
      unsigned char* Bp = stage + TILE_BYTES + wave * (NI * 1024) + lane * 16;
      const float* cf = reinterpret_cast<const float*>(stage + 2 * TILE_BYTES + wave * CW) + cch * 8;
      float c0[8], c1[8];
      uint4 raw[NI];
      if (COEF || SWISH) {
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[i] = lds_ld16(reinterpret_cast<const uint4*>(Ap + i * 1024));
      }
      if (COEF && affine) { ld8f(cf, c0); ld8f(cf + BK, c1); }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (COEF || SWISH) {
          float x[8];
          unpack8(raw[i], x);
          if (COEF && affine) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], c0[e], c1[e]);
          }
          if (SWISH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
          }
          if (COEF && gated) {
            float gt[8];
            ld8f(cf + (2 + ((gsel >> (2 * i)) & 3u)) * BK, gt);     // the row's image: an address, not 8 selects
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= gt[e];
          }
          v = pack8(x);
          if (ragged) { v.x = kok ? v.x : 0u; v.y = kok ? v.y : 0u; v.z = kok ? v.z : 0u; v.w = kok ? v.w : 0u; }
        }
        if ((COEF || SWISH) || !kok) lds_st16(reinterpret_cast<uint4*>(Ap + i * 1024), v);
        if (!kok || ((bzero >> i) & 1u)) lds_st16(reinterpret_cast<uint4*>(Bp + i * 1024), make_uint4(0, 0, 0, 0));
      }
    };
    // 64 x 64 per wave: acc[nj][mi] = D[i = output column within the 32-tile][j = row within the 32-tile]
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;
    // (all 16-deep sub-steps always: chunks past the reduction length are zero in both operands)
    auto mma = [&](const unsigned char* stage) {
      const unsigned char* As = stage;
      const unsigned char* Bs = stage + TILE_BYTES;
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const int slot = ((kk * 2 + h) ^ GE::swz(r)) * 16;
        const bf16x8 b0 = lds_ldf(reinterpret_cast<const bf16x8*>(Bs + (wj * 64 + r) * ROWB + slot));
        const bf16x8 b1 = lds_ldf(reinterpret_cast<const bf16x8*>(Bs + (wj * 64 + 32 + r) * ROWB + slot));
        const bf16x8 a0 = lds_ldf(reinterpret_cast<const bf16x8*>(As + (wm * 64 + r) * ROWB + slot));
        const bf16x8 a1 = lds_ldf(reinterpret_cast<const bf16x8*>(As + (wm * 64 + 32 + r) * ROWB + slot));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
      }
    };

    // prologue: three steps in flight, step 0 transformed
    issue(0, smem);
    if (nk > 1) issue(1, smem + STAGE_BYTES);
    if (nk > 2) issue(2, smem + 2 * STAGE_BYTES);
    if (nk > 2) wait_vm<2 * G>();
    else if (nk > 1) wait_vm<G>();
    else wait_vm<0>();
    transform_step(0, smem);
    barrier_lds();
    for (int p = 0; p < nk; ++p) {
      unsigned char* cur = smem + (p & 3) * STAGE_BYTES;
      if (p + 3 < nk) issue(p + 3, smem + ((p + 3) & 3) * STAGE_BYTES);
      if (p + 1 < nk) {
        if (p + 3 < nk) wait_vm<2 * G>();
        else if (p + 2 < nk) wait_vm<G>();
        else wait_vm<0>();
        transform_step(p + 1, smem + ((p + 1) & 3) * STAGE_BYTES);
      }
      mma(cur);
      barrier_lds();
    }

    // ---------------------------------------------------------------- epilogue through LDS (as k_big_gemm<false, false>)
    const int r = lane & 31, h = lane >> 5;
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
    __syncthreads();                     // the next row tile's DMAs overwrite the C tile / the sums
  }
  if (want_stats && tid < BJ && j0 + tid < a.J) {
    float* dst = a.stat_partials + (size_t)grp * 2 * a.J;
    dst[j0 + tid] = tot1;
    dst[a.J + j0 + tid] = tot2;
  }
}

template <int BK, bool COEF, bool SWISH> int launch(const Args& a, int grid, hipStream_t st) {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_fwd<BK, COEF, SWISH>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, Geo<BK>::SMEM_BYTES) == hipSuccess;
  if (!ok) return 0;
  edet_launch(k_wide_fwd<BK, COEF, SWISH>, dim3(grid), dim3(THREADS), Geo<BK>::SMEM_BYTES, st, a);
  return 1;
}
template <int BK> int launch_bk(const Args& a, int grid, bool coef, bool sw, hipStream_t st) {
  if (coef && sw) return launch<BK, true, true>(a, grid, st);
  if (coef) return launch<BK, true, false>(a, grid, st);
  if (sw) return launch<BK, false, true>(a, grid, st);
  return launch<BK, false, false>(a, grid, st);
}

}  // namespace pwg

// return 1 = handled, 0 = shape outside the envelope (the caller goes on to k_big_gemm), < 0 = error.
// Envelope: swish / linear views; with an SE gate at least 43 pixels per image (a row tile touches at most four images);
// reduction length a multiple of 8 and at least 2 x 64 (shorter reductions have nothing to prefetch).
int pwg_try_fwd(const edet_tview_t* in, const void* wt, int ldw, const float* bias, void* out, int cout,
                int ldo, float* stat_partials, int* nparts_out, int tpw, hipStream_t st) {
  using namespace pwg;
  const int K = in->c, N = cout;
  if (K % 8 != 0 || in->ld % 8 != 0 || ldw % 8 != 0 || ldo % 8 != 0 || ldo < (N + 7) / 8 * 8) return 0;
  if (in->act > EDET_ACT_SWISH || K < 128) return 0;
  if (in->gate && (BM - 1) / (in->h * in->w) + 2 > Geo<32>::NG) return 0;      // images a row tile may touch
  Args a;
  memset(&a, 0, sizeof(a));
  a.tv = *in;
  a.Bm = reinterpret_cast<const bf16_t*>(wt); a.ldb = ldw;
  a.M = in->n * in->h * in->w; a.R = K; a.J = N; a.hw = in->h * in->w;
  a.bias = bias; a.out = reinterpret_cast<bf16_t*>(out); a.ldo = ldo; a.stat_partials = stat_partials;
  a.ntm = (a.M + BM - 1) / BM; a.ntj = (N + BJ - 1) / BJ;
  a.tpw = tpw;
  a.ngrp = (a.ntm + a.tpw - 1) / a.tpw;
  const int grid = (a.ngrp + 7) / 8 * 8 * a.ntj;
  const bool coef = in->scale != nullptr || in->gate != nullptr, sw = in->act == EDET_ACT_SWISH;
  const int rc = launch_bk<32>(a, grid, coef, sw, st);
  if (rc != 1) return rc;
  if (nparts_out) *nparts_out = a.ngrp;
  EDET_LAUNCH_CHECK("edet_pw_fwd(glds)");
  return 1;
}
