// Shared device helpers for the gfx950 EfficientDet kernels (wave64, NHWC, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/edet_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ---------------------------------------------------------------- error handling
void edet_set_error(const char* fmt, ...);
#define EDET_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      edet_set_error(__VA_ARGS__);       \
      return -1;                         \
    }                                    \
  } while (0)
#define EDET_LAUNCH_CHECK(name)                                          \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      edet_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return -2;                                                         \
    }                                                                    \
  } while (0)

// ---------------------------------------------------------------- kernel launches
// Every kernel of the library is launched through edet_launch, so that the debug launch log
// (edet_debug_launch_log / edet_debug_launch_names in include/edet_hip.h) can name the kernel symbols a run
// used: the parity tests assert that every symbol of the full-size benchmark step is also launched by a
// test that checks results against the oracle.  With the log off the cost is one predictable branch.
#include <tuple>
#include <utility>
extern int g_edet_launch_log_on;
void edet_log_launch(const void* host_fn);
template <typename Tuple, size_t... I>
inline hipError_t edet_launch_impl(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, Tuple& params,
                                   std::index_sequence<I...>) {
  void* ptrs[] = {static_cast<void*>(&std::get<I>(params))..., nullptr};
  return hipLaunchKernel(fn, grid, block, ptrs, lds, st);
}
// kern<<<grid, block, lds, st>>>(args...) with the arguments converted to the kernel's parameter types
template <typename... KArgs, typename... Args>
inline void edet_launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "edet_launch: wrong number of kernel arguments");
  const void* fn = reinterpret_cast<const void*>(kern);
  if (g_edet_launch_log_on) edet_log_launch(fn);
  std::tuple<typename std::decay<KArgs>::type...> params(static_cast<KArgs>(args)...);
  (void)edet_launch_impl(fn, grid, block, lds, st, params, std::index_sequence_for<KArgs...>{});
}

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16: gfx950 has v_cvt_pk_bf16_f32, which the compiler selects for a
// plain float -> __bf16 conversion (one instruction per two elements)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// ---------------------------------------------------------------- 8-element vector I/O
// p must be 16-byte aligned for bf16 (8 elems) and fp32 (2 x float4).
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) {
  uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) {
  uint4 a;
  a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]);
  a.z = pack2bf(v[4], v[5]); a.w = pack2bf(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}
__device__ __forceinline__ void loadf8(const float* p, float v[8]) { load8<float>(p, v); }

// ---------------------------------------------------------------- 4-element vector I/O
// 8-byte aligned for bf16, 16-byte aligned for fp32.
template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
  uint2 a = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
  uint2 a;
  a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = a;
}

// ---------------------------------------------------------------- activations
// v_exp_f32 + v_rcp_f32 (1 ulp each); a full-precision divide would cost ~10 more VALU instructions
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx x*sigmoid(x) = s * (1 + x * (1 - s))
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// The other activation types of utils.activation_fn (act > EDET_ACT_SWISH): relu, relu6, hswish, mish, srelu.
// Derivatives as TensorFlow's gradient kernels define them at the kinks (ReluGrad: x > 0; Relu6Grad: 0 < x < 6).
// mish: tanh(softplus(z)) = w / (w + 2) with w = e^z (e^z + 2) (exact identity; z > 20: 1 to fp32 precision).
// srelu (utils.srelu_fn, utils.py:25-31): beta = (20^2)^2 -- the reference makes its `srelu_beta` variable afresh, at 20.0,
// in every call, so it is a constant of the graph, not a trained value.
constexpr float SRELU_BETA = 160000.f;
__device__ __forceinline__ float act_other_(int act, float z) {
  if (act == EDET_ACT_RELU) return fmaxf(z, 0.f);
  if (act == EDET_ACT_RELU6) return fminf(fmaxf(z, 0.f), 6.f);
  if (act == EDET_ACT_HSWISH) return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) / 6.f;
  if (act == EDET_ACT_MISH) {
    if (z > 20.f) return z;
    const float n = __expf(z), w = n * (n + 2.f);
    return z * w / (w + 2.f);
  }
  return z > 0.f ? z - __logf(fmaf(SRELU_BETA, z, 1.f)) * (1.f / SRELU_BETA) : 0.f;
}
__device__ __forceinline__ float act_other_grad_(int act, float z) {
  if (act == EDET_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == EDET_ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
  if (act == EDET_ACT_HSWISH) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) / 6.f);
  if (act == EDET_ACT_MISH) {
    if (z > 20.f) return 1.f;
    const float n = __expf(z), w = n * (n + 2.f), t = w / (w + 2.f);
    return fmaf(z * (1.f - t * t), n / (1.f + n), t);          // t + z (1 - t^2) sigmoid(z)
  }
  return z > 0.f ? 1.f - 1.f / fmaf(SRELU_BETA, z, 1.f) : 0.f;
}
// any activation code (kernel-uniform): value and derivative
__device__ __forceinline__ float act_apply_(int act, float z) {
  return act == EDET_ACT_SWISH ? swishf_(z) : (act == EDET_ACT_NONE ? z : act_other_(act, z));
}
__device__ __forceinline__ float act_grad_(int act, float z) {
  return act == EDET_ACT_SWISH ? swish_gradf_(z) : (act == EDET_ACT_NONE ? 1.f : act_other_grad_(act, z));
}

// Activated-view element transform for 8 consecutive channels starting at c0.
// img = image index of the row (only used when gate != NULL).
struct ViewCoef {
  float scale[8], shift[8];
};
__device__ __forceinline__ void view_load_coef(const edet_tview_t& v, int c0, ViewCoef& k) {
  if (v.scale) {
    loadf8(v.scale + c0, k.scale);
    loadf8(v.shift + c0, k.shift);
  }
}
// x (raw) -> z (pre-activation) -> value
__device__ __forceinline__ void view_apply(const edet_tview_t& v, const ViewCoef& k, int c0,
                                           int img, float x[8]) {
  if (v.scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], k.scale[e], k.shift[e]);
  }
  if (v.act == EDET_ACT_SWISH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = swishf_(x[e]);
  } else if (v.act > EDET_ACT_SWISH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = act_other_(v.act, x[e]);
  }
  if (v.gate) {
    float g[8];
    loadf8(v.gate + (size_t)img * v.c + c0, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] *= g[e];
  }
}

struct GradCoef {
  float a[8], b[8], cc[8];
};
__device__ __forceinline__ void grad_load_coef(const edet_gview_t& g, int c0, GradCoef& k) {
  if (g.a) {
    loadf8(g.a + c0, k.a);
    loadf8(g.b + c0, k.b);
    loadf8(g.cc + c0, k.cc);
  }
}
// loads dy for 8 channels at element offset `off` (= row*ld + c0)
template <typename T>
__device__ __forceinline__ void grad_load(const edet_gview_t& g, const GradCoef& k, size_t off,
                                          float dy[8]) {
  load8<T>(reinterpret_cast<const T*>(g.dz) + off, dy);
  if (g.a) {
    float y[8];
    load8<T>(reinterpret_cast<const T*>(g.y) + off, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) dy[e] = fmaf(k.a[e], dy[e], fmaf(k.b[e], y[e], k.cc[e]));
  }
}

// ---------------------------------------------------------------- ordered (run-to-run identical) sums
// Sum of v[] over the lanes of a wave that share lane % group (group: a power of two <= 64), by an xor butterfly:
// a fixed tree, so the same bits on every run; afterwards every lane of a class holds the class total.  All 64 lanes
// must be active at the call.  Used by the fp32 / generic kernels instead of LDS atomics; the waves of a workgroup then
// add their totals into LDS one wave after the other (wave order), and workgroups hand partial rows to
// edet_reduce_partials -- no floating-point atomics anywhere on the way.
template <int NV>
__device__ __forceinline__ void wave_group_sum(float (&v)[NV], int group) {
  for (int off = group; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] += __shfl_xor(v[e], off, 64);
  }
}

// Internal bit of edet_bwd_epi_t::flags (set by the bf16 dispatch of edet_pw_bwd / edet_pw_bwd_data, never by callers): the
// SE gate-gradient sums of this data-gradient launch are formed AFTERWARDS from the stored gradient by k_gate_sums (one
// writer per element, a fixed order) -- the kernel stores d as it is and adds nothing to dgate itself.  r06: the wide
// two-kernel paths (pw_big.hip, pw_stream.hip) added their sums with floating-point atomics, the last ones of the bf16
// training step (efficientdet-d7x: 1344 -> 224 / 960 -> 160 at 96 x 96).
#define EDET_EPI_GATE_SUMS_LATER 0x40000000

// TF 'SAME' geometry
__host__ __device__ inline int same_out(int in, int s) { return (in + s - 1) / s; }
__host__ __device__ inline int same_pad_before(int in, int k, int s) {
  int out = (in + s - 1) / s;
  int total = (out - 1) * s + k - in;
  if (total < 0) total = 0;
  return total / 2;
}

// dst[i] += sum_p ws[p*n + i]  (p < P): 16 elements x 16 partial-row slices per workgroup (bn_se.hip)
int edet_reduce_partials(const float* ws, int P, int64_t n, float* dst, hipStream_t st);
int edet_reduce_partials_set(const float* ws, int P, int64_t n, float* dst, hipStream_t st);
// two destinations: dst_a[i] += column i for i < n_a (dst_a may be NULL), dst_b[i - n_a] += column i for n_a <= i < n
int edet_reduce_partials2(const float* ws, int P, int64_t n, float* dst_a, int64_t n_a, float* dst_b, hipStream_t st);

// workgroups of kernel `fn` (block size `threads`, `lds` bytes of dynamic LDS) the device holds at once; 0 = unknown
int edet_resident_wgs(const void* fn, int threads, size_t lds);

static inline hipStream_t to_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
