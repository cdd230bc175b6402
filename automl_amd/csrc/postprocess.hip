// Detection post-processing on device (gfx950): SURVEY.md 8(f) row 1.
//
//   edet_pre_nms     tf2/postprocess.py pre_nms :120-157 with topk_class_boxes' per-anchor max class (:104-115):
//                    merge of the level outputs (:67-79), argmax / max over the classes, sigmoid, box decoding
//                    against the anchors (tf2/anchors.py:30-58).  One pass over the logits: HBM-bound
//                    (B * N * C elements read once, 24 bytes written per anchor).
//   edet_pre_nms_topk  the max_nms_inputs > 0 branch (:90-103): per-image top-k over the flattened (anchor, class)
//                    logits by a three-level radix select on the float keys + an LDS bitonic sort of the survivors.
//   edet_nms         greedy (soft) non-maximum suppression, one workgroup per segment (image, or image x class):
//                    tf.raw_ops.NonMaxSuppressionV5 semantics (postprocess.nms :160-206) or nms_np.py's
//                    (hard_nms :84-120, soft_nms :123-184); the per-class lists are merged per image the way
//                    postprocess.per_class_nms :447-460 / nms_np.per_class_nms :251-253 do (top max_output_size by
//                    score).  Eager formulation: after every selection the scores of all live candidates are
//                    decayed and the next maximum is found in the same pass.
//   edet_nms_gather  boxes / classes of the selected candidates, padding rows, clip_boxes :61-64, image scales.
//
// These are byte-shuffling, latency-bound kernels on small data (<= 76,725 candidates per image for D0-640); the
// design rule is one coalesced pass per step, scores and boxes of a segment contiguous in a workspace slice.
#include <math.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int TOPK_BLOCKS = 1024;   // workgroups (contiguous chunks) per image of the top-k passes

struct PreArgs {
  const void* cls[MAX_LEVELS];
  const void* box[MAX_LEVELS];
  int aoff[MAX_LEVELS + 1];     // first anchor of each level
  int lanch[MAX_LEVELS];        // anchors per image of each level
  int nlevels, batch, C, N;
  const float* anchors;         // [N][4] ymin, xmin, ymax, xmax
  float* boxes;
  float* scores;
  int* classes;
};

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// anchors.decode_box_outputs, op for op (no fused multiply-add: the reference graph rounds every product)
__device__ __forceinline__ float4 decode_box(const float4 code, const float4 an) {
  const float yc_a = __fmul_rn(__fadd_rn(an.x, an.z), 0.5f);
  const float xc_a = __fmul_rn(__fadd_rn(an.y, an.w), 0.5f);
  const float ha = __fsub_rn(an.z, an.x), wa = __fsub_rn(an.w, an.y);
  const float w = __fmul_rn(expf(code.w), wa), h = __fmul_rn(expf(code.z), ha);
  const float yc = __fadd_rn(__fmul_rn(code.x, ha), yc_a);
  const float xc = __fadd_rn(__fmul_rn(code.y, wa), xc_a);
  const float hh = __fmul_rn(h, 0.5f), hw = __fmul_rn(w, 0.5f);
  return make_float4(__fsub_rn(yc, hh), __fsub_rn(xc, hw), __fadd_rn(yc, hh), __fadd_rn(xc, hw));
}

template <typename T>
__device__ __forceinline__ float4 load_code(const T* p) {
  return make_float4(to_f<T>(p[0]), to_f<T>(p[1]), to_f<T>(p[2]), to_f<T>(p[3]));
}

// One wave per 64 consecutive anchors of one level: their 64 * C logits are contiguous in memory, so the wave copies
// them into LDS with whole-wave coalesced loads (VB bytes per lane: 16 when every block start is 16-byte aligned,
// else 4 or the element size) and every lane then scans its own row (row stride C elements: odd word strides such
// as C = 90 bf16 = 45 words are bank-conflict free).  Block y = image, block x = 64-anchor group of a level.
template <typename T, int VB>
__global__ __launch_bounds__(64) void k_pre_nms(const PreArgs a, const int* __restrict__ group_level,
                                                 const int* __restrict__ group_first) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x, b = blockIdx.y;
  const int l = group_level[blockIdx.x];
  const int a0 = group_first[blockIdx.x];                 // first anchor of the group inside its level
  const int rows = min(64, a.lanch[l] - a0);
  const size_t local0 = (size_t)b * a.lanch[l] + a0;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(a.cls[l]) + local0 * a.C * sizeof(T);
  const int nbytes = rows * a.C * (int)sizeof(T);
  if (VB == 16) {
    for (int o = lane * 16; o < nbytes; o += 64 * 16)     // nbytes % 16 == 0 is part of the VB == 16 contract
      *reinterpret_cast<uint4*>(smem + o) = *reinterpret_cast<const uint4*>(src + o);
  } else if (VB == 4) {
    for (int o = lane * 4; o < nbytes; o += 64 * 4)
      *reinterpret_cast<uint32_t*>(smem + o) = *reinterpret_cast<const uint32_t*>(src + o);
  } else {
    for (int o = lane * (int)sizeof(T); o < nbytes; o += 64 * (int)sizeof(T))
      *reinterpret_cast<T*>(smem + o) = *reinterpret_cast<const T*>(src + o);
  }
  __syncthreads();
  if (lane >= rows) return;
  const T* lg = reinterpret_cast<const T*>(smem) + (size_t)lane * a.C;
  float best = to_f<T>(lg[0]);
  int arg = 0;
  for (int c = 1; c < a.C; ++c) {       // tf.math.argmax: the first maximum
    const float v = to_f<T>(lg[c]);
    if (v > best) { best = v; arg = c; }
  }
  const int n = a.aoff[l] + a0 + lane;
  const float4 code = load_code<T>(reinterpret_cast<const T*>(a.box[l]) + (local0 + lane) * 4);
  const float4 an = *reinterpret_cast<const float4*>(a.anchors + (size_t)n * 4);
  const size_t o = (size_t)b * a.N + n;
  *reinterpret_cast<float4*>(a.boxes + o * 4) = decode_box(code, an);
  a.scores[o] = sigmoid_exact(best);
  a.classes[o] = arg;
}

// ------------------------------------------------------------------------------------------------ top-k branch
// order-preserving key of a float: larger float <=> larger unsigned key
__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct TopkArgs {
  PreArgs p;
  int K;                 // max_nms_inputs
  int64_t total;         // N * C flat logits per image
  uint32_t* hist;        // [B][4096]
  uint32_t* state;       // [B][4]: prefix key, prefix mask, still-needed count, (unused)
  int* sel_flat;         // [B][Kpad] flat indices of the selected logits (unsorted, then sorted)
  float* sel_val;        // [B][Kpad]
  uint32_t* counters;    // [B][2]: written "greater" entries
  uint32_t* eq_count;    // [B][blocks] keys equal to the threshold in each block's chunk
  uint32_t* eq_base;     // [B][blocks] exclusive prefix of eq_count
  int Kpad;
  int shift, bits;       // radix digit of this pass
};

template <typename T>
__device__ __forceinline__ float flat_logit(const PreArgs& a, int b, int64_t flat, int* anchor, int* cls) {
  const int n = (int)(flat / a.C);
  const int c = (int)(flat - (int64_t)n * a.C);
  int l = 0;
  while (l + 1 < a.nlevels && n >= a.aoff[l + 1]) ++l;
  const size_t local = (size_t)b * a.lanch[l] + (n - a.aoff[l]);
  *anchor = n;
  *cls = c;
  return to_f<T>(reinterpret_cast<const T*>(a.cls[l])[local * a.C + c]);
}

// 8 consecutive flat logits starting at f0 (a multiple of 8): the logits of one level are contiguous per image, so
// a unit that lies inside a level and is 16-byte aligned is one (bf16) or two (fp32) vector loads
template <typename T>
__device__ __forceinline__ int load_unit(const PreArgs& a, int b, int64_t f0, int64_t total, float v[8]) {
  const int valid = (int)min((int64_t)8, total - f0);
  int l = 0;
  while (l + 1 < a.nlevels && f0 >= (int64_t)a.aoff[l + 1] * a.C) ++l;
  const int64_t lo = (int64_t)a.aoff[l] * a.C, hi = (int64_t)a.aoff[l + 1] * a.C;
  const T* p = reinterpret_cast<const T*>(a.cls[l]) + ((size_t)b * a.lanch[l] * a.C + (size_t)(f0 - lo));
  if (f0 + 8 <= hi && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    load8<T>(p, v);
    return 8;
  }
  for (int e = 0; e < 8; ++e) {
    v[e] = 0.f;
    if (e < valid) {
      int n, c;
      v[e] = flat_logit<T>(a, b, f0 + e, &n, &c);
    }
  }
  return valid;
}

template <typename T>
__global__ __launch_bounds__(256) void k_topk_hist(const TopkArgs a) {
  __shared__ uint32_t h[4096];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t prefix = a.state[b * 4 + 0], pmask = a.state[b * 4 + 1];
  const uint32_t dmask = (1u << a.bits) - 1u;
  for (int64_t f0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; f0 < a.total; f0 += (int64_t)gridDim.x * 2048) {
    float v[8];
    const int valid = load_unit<T>(a.p, b, f0, a.total, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t key = float_key(v[e]);
      if (e < valid && (key & pmask) == prefix) atomicAdd(&h[(key >> a.shift) & dmask], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256)
    if (h[i]) atomicAdd(&a.hist[(size_t)b * 4096 + i], h[i]);
}

// one workgroup per image: the digit bin that holds the K-th largest key among those matching the prefix
__global__ __launch_bounds__(256) void k_topk_pick(const TopkArgs a) {
  __shared__ uint32_t part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  uint32_t* h = a.hist + (size_t)b * 4096;
  const int nb = 1 << a.bits, per = (nb + 255) / 256;      // bins per thread: 16 (12-bit digit) or 1 (8-bit)
  uint32_t mine[16];
  uint32_t sum = 0;
  for (int i = 0; i < per; ++i) {
    const int d = t * per + i;
    mine[i] = d < nb ? h[d] : 0u;
    sum += mine[i];
    if (d < nb) h[d] = 0;                                   // ready for the next pass
  }
  part[t] = sum;
  __syncthreads();
  uint32_t above = 0;                                       // keys in the bins above this thread's range
  for (int u = t + 1; u < 256; ++u) above += part[u];
  const uint32_t need = a.state[b * 4 + 2];
  if (above < need && need <= above + sum) {               // exactly one thread
    uint32_t left = need - above;
    int d = t * per + per - 1;
    for (int i = per - 1; i > 0; --i, --d) {
      if (mine[i] >= left) break;
      left -= mine[i];
    }
    a.state[b * 4 + 0] |= (uint32_t)d << a.shift;
    a.state[b * 4 + 1] |= ((1u << a.bits) - 1u) << a.shift;
    a.state[b * 4 + 2] = left;      // how many keys EQUAL to the final threshold are still to be taken
  }
}

__global__ void k_topk_init(const TopkArgs a) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) a.hist[(size_t)b * 4096 + i] = 0;
  if (threadIdx.x == 0) {
    a.state[b * 4 + 0] = 0; a.state[b * 4 + 1] = 0; a.state[b * 4 + 2] = (uint32_t)a.K; a.state[b * 4 + 3] = 0;
    a.counters[b * 2 + 0] = 0; a.counters[b * 2 + 1] = 0;
  }
}

// Keys above the threshold go to the front part of the list (any order); keys EQUAL to it are taken in ascending
// flat-index order (tf.math.top_k keeps the lower index among ties, and bf16 logits tie often): every block owns
// a contiguous chunk of the flat range, counts its equal keys, a prefix over the blocks ranks them.
template <typename T, bool WRITE_EQUAL>
__global__ __launch_bounds__(256) void k_topk_collect(const TopkArgs a) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t run;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t thr = a.state[b * 4 + 0], need = a.state[b * 4 + 2];
  // contiguous chunk per block, a multiple of the 2048 logits one block iteration covers
  const int64_t chunk = ((a.total + gridDim.x - 1) / gridDim.x + 2047) / 2048 * 2048;
  const int64_t f_begin = (int64_t)blockIdx.x * chunk, f_end = min(a.total, f_begin + chunk);
  const size_t bslot = (size_t)b * gridDim.x + blockIdx.x;
  uint32_t eq_base = 0, first = 0;
  if (WRITE_EQUAL) {
    eq_base = a.eq_base[bslot];
    first = a.counters[b * 2 + 0];
    if (eq_base >= need || a.eq_count[bslot] == 0) return;
  }
  if (tid == 0) run = 0;
  __syncthreads();
  for (int64_t it = f_begin; it < f_end; it += 2048) {
    const int64_t f0 = it + (int64_t)tid * 8;
    float v[8];
    int valid = 0;
    if (f0 < f_end) valid = min(load_unit<T>(a.p, b, f0, a.total, v), (int)min((int64_t)8, f_end - f0));
    uint32_t neq = 0, ngt = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e < valid) {
        const uint32_t key = float_key(v[e]);
        neq += key == thr;
        ngt += key > thr;
      }
    }
    if (!WRITE_EQUAL && ngt) {
      uint32_t slot = atomicAdd(&a.counters[b * 2 + 0], ngt);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e < valid && float_key(v[e]) > thr) {
          a.sel_flat[(size_t)b * a.Kpad + slot] = (int)(f0 + e);
          a.sel_val[(size_t)b * a.Kpad + slot] = v[e];
          ++slot;
        }
      }
    }
    // exclusive prefix of neq over the block's threads (thread order = flat order)
    uint32_t inc = neq;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = run + inc - neq, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) pre += wsum[w];
      tot += wsum[w];
    }
    if (WRITE_EQUAL && neq) {
      uint32_t rank = eq_base + pre;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e < valid && float_key(v[e]) == thr) {
          if (rank < need) {
            a.sel_flat[(size_t)b * a.Kpad + first + rank] = (int)(f0 + e);
            a.sel_val[(size_t)b * a.Kpad + first + rank] = v[e];
          }
          ++rank;
        }
      }
    }
    __syncthreads();
    if (tid == 0) run += tot;
    __syncthreads();
  }
  if (!WRITE_EQUAL && tid == 0) a.eq_count[bslot] = run;
}

__global__ void k_topk_eq_prefix(const TopkArgs a, int blocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.p.batch) return;
  uint32_t acc = 0;
  for (int j = 0; j < blocks; ++j) {
    a.eq_base[(size_t)b * blocks + j] = acc;
    acc += a.eq_count[(size_t)b * blocks + j];
  }
}

// bitonic sort of the K selected (value descending, flat index ascending) in LDS, then decode
__global__ __launch_bounds__(1024) void k_topk_sort_decode(const TopkArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* sv = reinterpret_cast<float*>(smem);
  int* sf = reinterpret_cast<int*>(sv + a.Kpad);
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < a.Kpad; i += 1024) {
    const bool ok = i < a.K;
    sv[i] = ok ? a.sel_val[(size_t)b * a.Kpad + i] : -INFINITY;
    sf[i] = ok ? a.sel_flat[(size_t)b * a.Kpad + i] : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= a.Kpad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < a.Kpad; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const float vi = sv[i], vp = sv[p];
          const int fi = sf[i], fp = sf[p];
          const bool i_first = vi > vp || (vi == vp && fi < fp);      // i belongs before p in the final order
          const bool up = (i & k) == 0;
          if (up ? !i_first : i_first) {
            sv[i] = vp; sv[p] = vi;
            sf[i] = fp; sf[p] = fi;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < a.K; i += 1024) {
    const int64_t f = sf[i];
    const int n = (int)(f / a.p.C), c = (int)(f - (int64_t)n * a.p.C);
    int l = 0;
    while (l + 1 < a.p.nlevels && n >= a.p.aoff[l + 1]) ++l;
    const size_t o = (size_t)b * a.K + i;
    a.p.scores[o] = sigmoid_exact(sv[i]);
    a.p.classes[o] = c;
    reinterpret_cast<int*>(a.sel_flat)[(size_t)b * a.Kpad + i] = n;     // anchor index, for the box pass
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_topk_boxes(const TopkArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= a.K) return;
  const int n = a.sel_flat[(size_t)b * a.Kpad + i];
  int l = 0;
  while (l + 1 < a.p.nlevels && n >= a.p.aoff[l + 1]) ++l;
  const size_t local = (size_t)b * a.p.lanch[l] + (n - a.p.aoff[l]);
  const float4 code = load_code<T>(reinterpret_cast<const T*>(a.p.box[l]) + local * 4);
  const float4 an = *reinterpret_cast<const float4*>(a.p.anchors + (size_t)n * 4);
  *reinterpret_cast<float4*>(a.p.boxes + ((size_t)b * a.K + i) * 4) = decode_box(code, an);
}

// ------------------------------------------------------------------------------------------------ NMS
constexpr int NMS_THREADS = 512;
constexpr int NMS_WAVES = NMS_THREADS / 64;

struct NmsArgs {
  const float* boxes;      // [B][K][4]
  const float* scores;     // [B][K]
  const int* classes;      // [B][K]
  int K, S, M;
  int method;              // EDET_NMS_HARD / GAUSSIAN / LINEAR
  int convention;          // EDET_NMS_TF_V5 / EDET_NMS_NUMPY
  float iou_thr, score_thr, sigma;
  int* seg_offsets;        // [B][S+1]
  float4* w_box;
  float* w_score;
  int* w_idx;
  int* sel_index;          // [B][S][M]
  float* sel_score;
  int* sel_count;          // [B][S]
};

__global__ __launch_bounds__(NMS_THREADS) void k_class_offsets(const NmsArgs a) {
  extern __shared__ int hist[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < a.S; i += NMS_THREADS) hist[i] = 0;
  __syncthreads();
  for (int p = threadIdx.x; p < a.K; p += NMS_THREADS) {
    const int c = a.classes[(size_t)b * a.K + p];
    if (c >= 0 && c < a.S) atomicAdd(&hist[c], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    int* o = a.seg_offsets + (size_t)b * (a.S + 1);
    for (int s = 0; s < a.S; ++s) { o[s] = run; run += hist[s]; }
    o[a.S] = run;
  }
}

template <int CONV>
__device__ __forceinline__ float iou_of(const float4 p, const float4 q) {
  if (CONV == EDET_NMS_TF_V5) {
    // non_max_suppression_op.cc IOU: corners already ordered (min, min, max, max) when staged
    const float ai = (p.z - p.x) * (p.w - p.y), aj = (q.z - q.x) * (q.w - q.y);
    if (ai <= 0.f || aj <= 0.f) return 0.f;
    const float ih = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x), 0.f);
    const float iw = fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y), 0.f);
    const float inter = ih * iw;
    return inter / (ai + aj - inter);
  } else {
    // nms_np.py: pixel-inclusive extents (+1)
    const float ai = (p.z - p.x + 1.f) * (p.w - p.y + 1.f), aj = (q.z - q.x + 1.f) * (q.w - q.y + 1.f);
    const float ih = fmaxf(0.f, fminf(p.z, q.z) - fmaxf(p.x, q.x) + 1.f);
    const float iw = fmaxf(0.f, fminf(p.w, q.w) - fmaxf(p.y, q.y) + 1.f);
    const float inter = ih * iw;
    return inter / (ai + aj - inter);
  }
}

struct Best {
  float score;
  int pos;
};
__device__ __forceinline__ Best better(const Best x, const Best y) {
  return (y.score > x.score || (y.score == x.score && y.pos < x.pos)) ? y : x;
}

template <int CONV>
__global__ __launch_bounds__(NMS_THREADS) void k_nms_segment(const NmsArgs a) {
  __shared__ int wtot[NMS_WAVES];
  __shared__ float rs[NMS_WAVES];
  __shared__ int rp[NMS_WAVES];
  __shared__ int rf[NMS_WAVES];
  __shared__ float4 sel_box;
  __shared__ int sel_pos;
  const int s = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t img = (size_t)b * a.K;
  const bool per_class = a.S > 1;
  const int base = per_class ? a.seg_offsets[(size_t)b * (a.S + 1) + s] : 0;
  const bool soft = a.method != EDET_NMS_HARD;
  // hard_nms of nms_np.py has no score threshold at all; the V5 op filters `score > threshold` up front
  const bool filter_in = CONV == EDET_NMS_TF_V5;

  // ---- stage the segment's candidates, in ascending candidate order, into its workspace slice
  int count = 0;
  for (int p0 = 0; p0 < a.K; p0 += NMS_THREADS) {
    const int p = p0 + tid;
    bool take = false;
    float sc = 0.f;
    if (p < a.K) {
      sc = a.scores[img + p];
      take = (!per_class || a.classes[img + p] == s) && (!filter_in || sc > a.score_thr);
    }
    const unsigned long long m = __ballot(take);
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NMS_WAVES; ++w) {
      if (w < wave) pre += wtot[w];
      tot += wtot[w];
    }
    if (take) {
      const int pos = base + count + pre + __popcll(m & ((1ull << lane) - 1ull));
      float4 bx = *reinterpret_cast<const float4*>(a.boxes + (img + p) * 4);
      if (CONV == EDET_NMS_TF_V5)
        bx = make_float4(fminf(bx.x, bx.z), fminf(bx.y, bx.w), fmaxf(bx.x, bx.z), fmaxf(bx.y, bx.w));
      a.w_box[img + pos] = bx;
      a.w_score[img + pos] = sc;
      a.w_idx[img + pos] = p;
    }
    count += tot;
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();

  float4* wb = a.w_box + img + base;
  float* ws = a.w_score + img + base;
  const size_t so = ((size_t)b * a.S + s) * a.M;
  int nsel = 0;
  bool have_sel = false;
  while (true) {
    // one pass: decay every live candidate by the last selection (if any), track the maximum
    Best mine = {-INFINITY, 0x7fffffff};
    int first_live = 0x7fffffff;          // lowest live slot (nms_np: the row the maximum is swapped with)
    const float4 sb = sel_box;
    for (int i = tid; i < count; i += NMS_THREADS) {
      float sc = ws[i];
      if (sc == -INFINITY) continue;
      if (have_sel) {
        const float iou = iou_of<CONV>(wb[i], sb);
        if (!soft) {
          if (iou > a.iou_thr) sc = -INFINITY;
        } else {
          float w;
          if (a.method == EDET_NMS_LINEAR) w = iou > a.iou_thr ? 1.f - iou : 1.f;
          else if (CONV == EDET_NMS_TF_V5) w = expf((-0.5f / a.sigma) * (iou * iou));
          else w = expf(-(iou * iou) / a.sigma);
          sc *= w;
          const bool dead = CONV == EDET_NMS_TF_V5 ? sc <= a.score_thr : sc < a.score_thr;
          if (dead) sc = -INFINITY;
        }
        ws[i] = sc;
      }
      if (sc != -INFINITY) {
        mine = better(mine, Best{sc, i});
        first_live = min(first_live, i);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o;
      o.score = __shfl_xor(mine.score, off);
      o.pos = __shfl_xor(mine.pos, off);
      mine = better(mine, o);
      first_live = min(first_live, __shfl_xor(first_live, off));
    }
    if (lane == 0) { rs[wave] = mine.score; rp[wave] = mine.pos; rf[wave] = first_live; }
    __syncthreads();
    if (tid == 0) {
      Best g = {rs[0], rp[0]};
      int fl = rf[0];
      for (int w = 1; w < NMS_WAVES; ++w) { g = better(g, Best{rs[w], rp[w]}); fl = min(fl, rf[w]); }
      sel_pos = g.score == -INFINITY ? -1 : g.pos;
      if (sel_pos >= 0) {
        a.sel_index[so + nsel] = a.w_idx[img + base + g.pos];
        a.sel_score[so + nsel] = g.score;
        sel_box = wb[g.pos];
        ws[g.pos] = -INFINITY;
        if (CONV == EDET_NMS_NUMPY && soft && fl != g.pos) {
          // soft_nms swaps the maximum with row 0 (nms_np.py:158) before dropping it: the former row 0 now sits
          // where the maximum was, which decides later ties of np.argmax (first maximum in array order)
          wb[g.pos] = wb[fl];
          ws[g.pos] = ws[fl];
          a.w_idx[img + base + g.pos] = a.w_idx[img + base + fl];
          ws[fl] = -INFINITY;
        }
      }
    }
    __syncthreads();
    if (sel_pos < 0) break;
    ++nsel;
    have_sel = true;
    if (nsel == a.M) break;
  }
  if (tid == 0) a.sel_count[(size_t)b * a.S + s] = nsel;
}

struct MergeArgs {
  const int* sel_index;
  const float* sel_score;
  const int* sel_count;
  int S, M;
  int* out_index;      // [B][M]: candidate index, -1 = padding row
  float* out_score;
  int* out_valid;      // [B]
};

// top M of the concatenated per-segment lists by (score descending, position ascending)
__global__ __launch_bounds__(NMS_THREADS) void k_nms_merge(const MergeArgs a) {
  extern __shared__ unsigned char taken[];      // [S * M]
  __shared__ float rs[NMS_WAVES];
  __shared__ int rp[NMS_WAVES];
  __shared__ int pick;
  __shared__ int total_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int E = a.S * a.M;
  const int* cnt = a.sel_count + (size_t)b * a.S;
  const float* sc = a.sel_score + (size_t)b * E;
  for (int e = tid; e < E; e += NMS_THREADS) taken[e] = (e % a.M) >= cnt[e / a.M];
  if (tid == 0) {
    int t = 0;
    for (int s = 0; s < a.S; ++s) t += cnt[s];
    total_s = t;
  }
  __syncthreads();
  for (int m = 0; m < a.M; ++m) {
    Best mine = {-INFINITY, 0x7fffffff};
    for (int e = tid; e < E; e += NMS_THREADS)
      if (!taken[e]) mine = better(mine, Best{sc[e], e});
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o;
      o.score = __shfl_xor(mine.score, off);
      o.pos = __shfl_xor(mine.pos, off);
      mine = better(mine, o);
    }
    if (lane == 0) { rs[wave] = mine.score; rp[wave] = mine.pos; }
    __syncthreads();
    if (tid == 0) {
      Best g = {rs[0], rp[0]};
      for (int w = 1; w < NMS_WAVES; ++w) g = better(g, Best{rs[w], rp[w]});
      pick = g.pos == 0x7fffffff ? -1 : g.pos;
      if (pick >= 0) {
        taken[pick] = 1;
        a.out_index[(size_t)b * a.M + m] = a.sel_index[(size_t)b * E + pick];
        a.out_score[(size_t)b * a.M + m] = g.score;
      } else {
        a.out_index[(size_t)b * a.M + m] = -1;
        a.out_score[(size_t)b * a.M + m] = 0.f;
      }
    }
    __syncthreads();
  }
  if (tid == 0) a.out_valid[b] = min(a.M, total_s);
}

struct GatherArgs {
  const float* boxes;
  const int* classes;
  const int* out_index;
  const float* out_score;
  int K, M, pad_mode;
  float clip_h, clip_w;
  const float* scales;
  float* nms_boxes;
  float* nms_scores;
  float* nms_classes;
};

__global__ void k_nms_gather(const GatherArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (m >= a.M) return;
  const size_t o = (size_t)b * a.M + m;
  int idx = a.out_index[o];
  float score = a.out_score[o];
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  float cls = 0.f;
  if (idx < 0 && a.pad_mode == EDET_NMS_PAD_INDEX0) idx = 0;       // tf.gather(boxes, padded index 0)
  if (idx >= 0) {
    bx = *reinterpret_cast<const float4*>(a.boxes + ((size_t)b * a.K + idx) * 4);
    cls = (float)(a.classes[(size_t)b * a.K + idx] + 1);           // CLASS_OFFSET
  } else if (a.pad_mode == EDET_NMS_PAD_DUMMY) {
    score = -1e5f;                                                  // nms_np._DUMMY_DETECTION_SCORE
  }
  if (a.clip_h > 0.f) {
    bx.x = fminf(fmaxf(bx.x, 0.f), a.clip_h); bx.z = fminf(fmaxf(bx.z, 0.f), a.clip_h);
    bx.y = fminf(fmaxf(bx.y, 0.f), a.clip_w); bx.w = fminf(fmaxf(bx.w, 0.f), a.clip_w);
  }
  if (a.scales) {
    const float sc = a.scales[b];
    bx = make_float4(bx.x * sc, bx.y * sc, bx.z * sc, bx.w * sc);
  }
  *reinterpret_cast<float4*>(a.nms_boxes + o * 4) = bx;
  a.nms_scores[o] = score;
  a.nms_classes[o] = cls;
}

int fill_pre_args(PreArgs& a, const void* const* cls_levels, const void* const* box_levels, const int* level_pixels,
                  int nlevels, int batch, int anchors_per_pixel, int num_classes, const float* anchor_boxes,
                  float* boxes, float* scores, int* classes) {
  EDET_CHECK(nlevels >= 1 && nlevels <= MAX_LEVELS, "edet_pre_nms: %d levels (max %d)", nlevels, MAX_LEVELS);
  EDET_CHECK(batch >= 1 && anchors_per_pixel >= 1 && num_classes >= 1, "edet_pre_nms: bad sizes");
  int64_t run = 0;
  for (int l = 0; l < nlevels; ++l) {
    EDET_CHECK(cls_levels[l] && box_levels[l] && level_pixels[l] > 0, "edet_pre_nms: level %d is empty", l);
    a.cls[l] = cls_levels[l];
    a.box[l] = box_levels[l];
    a.aoff[l] = (int)run;
    a.lanch[l] = level_pixels[l] * anchors_per_pixel;
    run += a.lanch[l];
  }
  EDET_CHECK(run * (int64_t)num_classes < (int64_t)1 << 31, "edet_pre_nms: %lld x %d logits per image overflow int",
             (long long)run, num_classes);
  a.aoff[nlevels] = (int)run;
  a.nlevels = nlevels; a.batch = batch; a.C = num_classes; a.N = (int)run;
  a.anchors = anchor_boxes; a.boxes = boxes; a.scores = scores; a.classes = classes;
  return 0;
}

// device table [2][ngroups]: level and first anchor of every 64-anchor group, cached per level geometry
struct GroupTable {
  int nlevels;
  int lanch[MAX_LEVELS];
  int device;
  int* dev;
  int ngroups;
};
std::vector<GroupTable> g_tables;
std::mutex g_tables_mu;

int group_table(const PreArgs& a, const int** table, int* ngroups, hipStream_t st) {
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lock(g_tables_mu);
  for (const GroupTable& t : g_tables) {
    if (t.nlevels == a.nlevels && t.device == device && !memcmp(t.lanch, a.lanch, sizeof(int) * a.nlevels)) {
      *table = t.dev;
      *ngroups = t.ngroups;
      return 0;
    }
  }
  std::vector<int> lev, first;
  for (int l = 0; l < a.nlevels; ++l)
    for (int a0 = 0; a0 < a.lanch[l]; a0 += 64) { lev.push_back(l); first.push_back(a0); }
  GroupTable t;
  memset(&t, 0, sizeof(t));
  t.nlevels = a.nlevels;
  memcpy(t.lanch, a.lanch, sizeof(int) * a.nlevels);
  t.device = device;
  t.ngroups = (int)lev.size();
  lev.insert(lev.end(), first.begin(), first.end());
  if (hipMalloc(reinterpret_cast<void**>(&t.dev), lev.size() * sizeof(int)) != hipSuccess ||
      hipMemcpyAsync(t.dev, lev.data(), lev.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    edet_set_error("edet_pre_nms: group table upload failed");
    return -1;
  }
  g_tables.push_back(t);
  *table = t.dev;
  *ngroups = t.ngroups;
  return 0;
}

}  // namespace

extern "C" int edet_pre_nms(const void* const* cls_levels, const void* const* box_levels, const int* level_pixels,
                            int nlevels, int batch, int anchors_per_pixel, int num_classes,
                            const float* anchor_boxes, int dtype, float* boxes, float* scores, int* classes,
                            void* stream) {
  PreArgs a;
  if (fill_pre_args(a, cls_levels, box_levels, level_pixels, nlevels, batch, anchors_per_pixel, num_classes,
                    anchor_boxes, boxes, scores, classes)) return -1;
  EDET_CHECK(dtype == EDET_F32 || dtype == EDET_BF16, "edet_pre_nms: dtype %d", dtype);
  // 64-anchor groups per level; the (level, first anchor) table is a function of the level sizes only and is
  // cached on the device per geometry
  const int es = dtype == EDET_BF16 ? 2 : 4;
  int vb = 16;
  for (int l = 0; l < nlevels; ++l) {
    const int64_t img_bytes = (int64_t)a.lanch[l] * num_classes * es;     // start of image b = b * img_bytes
    const int64_t grp_bytes = (int64_t)64 * num_classes * es;
    const bool al16 = img_bytes % 16 == 0 && grp_bytes % 16 == 0 && ((uintptr_t)cls_levels[l]) % 16 == 0;
    const bool al4 = img_bytes % 4 == 0 && grp_bytes % 4 == 0 && ((uintptr_t)cls_levels[l]) % 4 == 0;
    if (!al16) vb = std::min(vb, al4 ? 4 : es);
  }
  const int* table = nullptr;
  int ngroups = 0;
  if (group_table(a, &table, &ngroups, to_stream(stream))) return -2;
  const dim3 grid(ngroups, batch);
  const size_t lds = (size_t)64 * num_classes * es + 16;
  EDET_CHECK(lds <= 160 * 1024, "edet_pre_nms: %d classes need %zu bytes of LDS per wave", num_classes, lds);
  hipStream_t st = to_stream(stream);
#define PRE_CASE(T, VB)                                                                                      \
  do {                                                                                                         \
    if (lds > 48 * 1024)                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pre_nms<T, VB>),                                     \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    edet_launch(k_pre_nms<T, VB>, grid, dim3(64), lds, st, a, table, table + ngroups);                                        \
  } while (0)
  if (dtype == EDET_BF16) {
    if (vb == 16) PRE_CASE(bf16_t, 16); else if (vb == 4) PRE_CASE(bf16_t, 4); else PRE_CASE(bf16_t, 2);
  } else {
    if (vb == 16) PRE_CASE(float, 16); else PRE_CASE(float, 4);
  }
#undef PRE_CASE
  EDET_LAUNCH_CHECK("edet_pre_nms");
  return 0;
}

extern "C" int edet_pre_nms_topk_workspace_bytes(int batch, int k, size_t* bytes) {
  EDET_CHECK(batch >= 1 && k >= 1 && k <= 8192, "edet_pre_nms_topk: k = %d (1..8192)", k);
  int kpad = 1;
  while (kpad < k) kpad <<= 1;
  *bytes = (size_t)batch * (4096 * 4 + 4 * 4 + 2 * 4 + (size_t)kpad * 8 + 2 * (size_t)TOPK_BLOCKS * 4) + 256;
  return 0;
}

extern "C" int edet_pre_nms_topk(const void* const* cls_levels, const void* const* box_levels,
                                 const int* level_pixels, int nlevels, int batch, int anchors_per_pixel,
                                 int num_classes, const float* anchor_boxes, int dtype, int k, void* workspace,
                                 size_t workspace_bytes, float* boxes, float* scores, int* classes, void* stream) {
  TopkArgs a;
  if (fill_pre_args(a.p, cls_levels, box_levels, level_pixels, nlevels, batch, anchors_per_pixel, num_classes,
                    anchor_boxes, boxes, scores, classes)) return -1;
  EDET_CHECK(dtype == EDET_F32 || dtype == EDET_BF16, "edet_pre_nms_topk: dtype %d", dtype);
  size_t need = 0;
  if (edet_pre_nms_topk_workspace_bytes(batch, k, &need)) return -1;
  EDET_CHECK(workspace && workspace_bytes >= need, "edet_pre_nms_topk: workspace %zu < %zu bytes", workspace_bytes, need);
  a.K = k;
  a.total = (int64_t)a.p.N * num_classes;
  EDET_CHECK(a.total >= k, "edet_pre_nms_topk: k = %d exceeds the %lld logits of an image", k, (long long)a.total);
  a.Kpad = 1;
  while (a.Kpad < k) a.Kpad <<= 1;
  hipStream_t st = to_stream(stream);
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  a.hist = reinterpret_cast<uint32_t*>(w); w += (size_t)batch * 4096 * 4;
  a.state = reinterpret_cast<uint32_t*>(w); w += (size_t)batch * 16;
  a.counters = reinterpret_cast<uint32_t*>(w); w += (size_t)batch * 8;
  a.sel_flat = reinterpret_cast<int*>(w); w += (size_t)batch * a.Kpad * 4;
  a.sel_val = reinterpret_cast<float*>(w); w += (size_t)batch * a.Kpad * 4;
  a.eq_count = reinterpret_cast<uint32_t*>(w); w += (size_t)batch * TOPK_BLOCKS * 4;
  a.eq_base = reinterpret_cast<uint32_t*>(w);
  edet_launch(k_topk_init, dim3(batch), dim3(256), 0, st, a);
  const int blocks = (int)std::min<int64_t>((a.total + 255) / 256, TOPK_BLOCKS);
  const dim3 grid(blocks, batch);
  const int shifts[3] = {20, 8, 0}, nbits[3] = {12, 12, 8};
  for (int pass = 0; pass < 3; ++pass) {
    a.shift = shifts[pass];
    a.bits = nbits[pass];
    if (dtype == EDET_BF16) edet_launch(k_topk_hist<bf16_t>, grid, dim3(256), 0, st, a);
    else edet_launch(k_topk_hist<float>, grid, dim3(256), 0, st, a);
    edet_launch(k_topk_pick, dim3(batch), dim3(256), 0, st, a);
  }
  if (dtype == EDET_BF16) edet_launch(k_topk_collect<bf16_t, false>, grid, dim3(256), 0, st, a);
  else edet_launch(k_topk_collect<float, false>, grid, dim3(256), 0, st, a);
  edet_launch(k_topk_eq_prefix, dim3(cdiv(batch, 64)), dim3(64), 0, st, a, blocks);
  if (dtype == EDET_BF16) edet_launch(k_topk_collect<bf16_t, true>, grid, dim3(256), 0, st, a);
  else edet_launch(k_topk_collect<float, true>, grid, dim3(256), 0, st, a);
  edet_launch(k_topk_sort_decode, dim3(batch), dim3(1024), (size_t)a.Kpad * 8, st, a);
  if (dtype == EDET_BF16) edet_launch(k_topk_boxes<bf16_t>, dim3(cdiv(k, 256), batch), dim3(256), 0, st, a);
  else edet_launch(k_topk_boxes<float>, dim3(cdiv(k, 256), batch), dim3(256), 0, st, a);
  EDET_LAUNCH_CHECK("edet_pre_nms_topk");
  return 0;
}

extern "C" int edet_nms_workspace_bytes(int batch, int n, int segments, int max_output_size, size_t* bytes) {
  EDET_CHECK(batch >= 1 && n >= 1 && segments >= 1 && max_output_size >= 1, "edet_nms: bad sizes");
  *bytes = (size_t)batch * n * (16 + 4 + 4) + (size_t)batch * (segments + 1) * 4 +
           (size_t)batch * segments * max_output_size * 8 + (size_t)batch * segments * 4 + 256;
  return 0;
}

extern "C" int edet_nms(const float* boxes, const float* scores, const int* classes, int batch, int n, int segments,
                        const edet_nms_cfg_t* cfg, void* workspace, size_t workspace_bytes, int* out_index,
                        float* out_score, int* out_valid, void* stream) {
  EDET_CHECK(boxes && scores && cfg && out_index && out_score && out_valid, "edet_nms: null argument");
  EDET_CHECK(segments == 1 || classes, "edet_nms: per-class suppression needs the class of every candidate");
  EDET_CHECK(cfg->method == EDET_NMS_HARD || cfg->method == EDET_NMS_GAUSSIAN ||
             (cfg->method == EDET_NMS_LINEAR && cfg->convention == EDET_NMS_NUMPY),
             "edet_nms: method %d is not defined for convention %d", cfg->method, cfg->convention);
  EDET_CHECK(cfg->convention == EDET_NMS_TF_V5 || cfg->convention == EDET_NMS_NUMPY, "edet_nms: convention %d",
             cfg->convention);
  EDET_CHECK(cfg->method == EDET_NMS_HARD || cfg->method == EDET_NMS_LINEAR || cfg->sigma > 0.f,
             "edet_nms: gaussian suppression needs sigma > 0");
  const int M = cfg->max_output_size;
  EDET_CHECK(M >= 1 && (size_t)segments * M <= 60000, "edet_nms: segments x max_output_size = %d x %d exceeds 60000",
             segments, M);
  size_t need = 0;
  if (edet_nms_workspace_bytes(batch, n, segments, M, &need)) return -1;
  EDET_CHECK(workspace && workspace_bytes >= need, "edet_nms: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t st = to_stream(stream);
  NmsArgs a;
  a.boxes = boxes; a.scores = scores; a.classes = classes;
  a.K = n; a.S = segments; a.M = M;
  a.method = cfg->method; a.convention = cfg->convention;
  a.iou_thr = cfg->iou_thresh; a.score_thr = cfg->score_thresh; a.sigma = cfg->sigma;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  a.w_box = reinterpret_cast<float4*>(w); w += (size_t)batch * n * 16;
  a.w_score = reinterpret_cast<float*>(w); w += (size_t)batch * n * 4;
  a.w_idx = reinterpret_cast<int*>(w); w += (size_t)batch * n * 4;
  a.seg_offsets = reinterpret_cast<int*>(w); w += (size_t)batch * (segments + 1) * 4;
  a.sel_index = reinterpret_cast<int*>(w); w += (size_t)batch * segments * M * 4;
  a.sel_score = reinterpret_cast<float*>(w); w += (size_t)batch * segments * M * 4;
  a.sel_count = reinterpret_cast<int*>(w);
  if (segments > 1) edet_launch(k_class_offsets, dim3(batch), dim3(NMS_THREADS), (size_t)segments * 4, st, a);
  const dim3 grid(segments, batch);
  if (cfg->convention == EDET_NMS_TF_V5) edet_launch(k_nms_segment<EDET_NMS_TF_V5>, grid, dim3(NMS_THREADS), 0, st, a);
  else edet_launch(k_nms_segment<EDET_NMS_NUMPY>, grid, dim3(NMS_THREADS), 0, st, a);
  MergeArgs m;
  m.sel_index = a.sel_index; m.sel_score = a.sel_score; m.sel_count = a.sel_count;
  m.S = segments; m.M = M;
  m.out_index = out_index; m.out_score = out_score; m.out_valid = out_valid;
  edet_launch(k_nms_merge, dim3(batch), dim3(NMS_THREADS), (size_t)segments * M, st, m);
  EDET_LAUNCH_CHECK("edet_nms");
  return 0;
}

extern "C" int edet_nms_gather(const float* boxes, const int* classes, const int* out_index, const float* out_score,
                               int batch, int n, int max_output_size, int pad_mode, float clip_h, float clip_w,
                               const float* image_scales, float* nms_boxes, float* nms_scores, float* nms_classes,
                               void* stream) {
  EDET_CHECK(boxes && classes && out_index && out_score && nms_boxes && nms_scores && nms_classes,
             "edet_nms_gather: null argument");
  EDET_CHECK(pad_mode == EDET_NMS_PAD_INDEX0 || pad_mode == EDET_NMS_PAD_ZERO || pad_mode == EDET_NMS_PAD_DUMMY,
             "edet_nms_gather: pad_mode %d", pad_mode);
  GatherArgs g;
  g.boxes = boxes; g.classes = classes; g.out_index = out_index; g.out_score = out_score;
  g.K = n; g.M = max_output_size; g.pad_mode = pad_mode;
  g.clip_h = clip_h; g.clip_w = clip_w; g.scales = image_scales;
  g.nms_boxes = nms_boxes; g.nms_scores = nms_scores; g.nms_classes = nms_classes;
  edet_launch(k_nms_gather, dim3(cdiv(max_output_size, 128), batch), dim3(128), 0, to_stream(stream), g);
  EDET_LAUNCH_CHECK("edet_nms_gather");
  return 0;
}
