// Detection loss (focal + Huber) forward and backward in one pass, and the fused optimizer
// (L2 regularisation, per-tensor + global gradient clipping, SGD momentum, EMA) over a flat
// fp32 parameter arena.
//
// Reference: efficientdet/tf2/train_lib.py:357-406 (FocalLoss), :409-437 (BoxLoss / Keras Huber),
// :493-604 (_detection_loss), :486-491 (_reg_l2_loss), :675-683 (clip + apply), :176-199 (optimizer).
#include "common.h"

namespace {

constexpr int THREADS = 256;

struct RowMap { int tpr, rpp; };
inline RowMap row_map_ld(int ld) {
  int nvec = ld / 8, tpr = 1;
  while (tpr < nvec && tpr < THREADS) tpr <<= 1;
  RowMap m; m.tpr = tpr; m.rpp = THREADS / tpr;
  return m;
}

// logits [positions][ld], channel j = anchor * num_classes + class.
// Per element (train_lib.py:357-406 with label_smoothing 0): u = +-x, 1 - p_t = sigmoid(u),
// ce = softplus(u); loss = alpha_t * sigmoid(u)^gamma * softplus(u) / normalizer.  One exp, one rcp,
// one log and one sqrt (gamma = 1.5) or exp2/log2 pair (general gamma) per logit; the anchor index is
// advanced incrementally along the 8-wide chunk, so there is one integer division per 16 bytes.
// LS: label smoothing (tf2/train_lib.py:400-402): the cross entropy is taken against y*(1-ls) + ls/2 while alpha and the
// modulating factor keep the hard label.  With u = -x for the positive class and x otherwise, ce_smoothed = softplus(u) -
// (ls/2) u in both cases, so d/du [sg^gamma (sp - h u)] = sg^gamma (gamma (1-sg) (sp - h u) + sg - h), h = ls/2.  A
// template parameter of the shared body: k_focal (ls = 0) keeps its instruction count (the kernel is VALU-bound) and
// its symbol; k_focal_ls is the smoothed one.
// Tail of the two loss kernels: the workgroup's loss sum and its per-channel bias-gradient sums.  r04: combined in a fixed
// order (wave shuffles, waves in order, row-lanes in order through LDS -- no LDS atomics); with a partial buffer the
// workgroup writes its row [nch | 1] there and edet_reduce_partials2 adds the rows in order (the same loss and bias gradient on
// every run); without one the kernel is launched as ONE workgroup, which adds into the destinations itself -- no atomics.
// LDS: scr[THREADS * 8] floats (dynamic).
__device__ __forceinline__ void loss_tail(float loss_acc, const float (&db)[8], bool ok, int nch, const RowMap& m, float* scr,
                                          float* part, float* sum_dst, float* dbias) {
  __shared__ float wsum[THREADS / 64];
  const int tid = threadIdx.x;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  const int width = m.tpr * 8;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) loss_acc += __shfl_down(loss_acc, off, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = loss_acc;
#pragma unroll
  for (int e = 0; e < 8; ++e) scr[rr * width + cv * 8 + e] = ok ? db[e] : 0.f;
  __syncthreads();
  float* row = part ? part + (size_t)blockIdx.x * (1 + nch) : nullptr;      // [bias gradient (nch) | loss]
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) t += wsum[w];
    if (row) row[nch] = t; else *sum_dst += t;          // (no partial buffer: the kernel runs as ONE workgroup)
  }
  if (dbias || row) {
    for (int i = tid; i < nch; i += THREADS) {
      float t = 0.f;
      for (int r = 0; r < m.rpp; ++r) t += scr[r * width + i];
      if (row) row[i] = t; else dbias[i] += t;
    }
  }
}

template <typename T, bool G15, bool LS>
__device__ __forceinline__ void focal_body(const T* __restrict__ logits, int ld,
                                                  const int32_t* __restrict__ tgt, int64_t positions,
                                                  int na, int nc, float alpha, float gamma, float inv_norm_h,
                                                  const float* __restrict__ norm_scale,
                                                  T* __restrict__ dlogits, float* dbias, float* sums, float* part, RowMap m,
                                                  float half_ls) {
  const float inv_norm = norm_scale ? inv_norm_h * norm_scale[0] : inv_norm_h;
  const int tid = threadIdx.x;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  const int j0 = cv * 8;
  const int nch = na * nc;
  const bool ok = j0 < ld;
  const int a0 = j0 / nc, k0 = j0 - a0 * nc;
  float loss_acc = 0.f, db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) db[e] = 0.f;
  if (ok && nc < 8) {
    // fewer than 8 classes: an 8-element chunk can span more than two anchors -- the simple form, one row at a time,
    // the anchor's target fetched when the class index wraps
    for (int64_t p = (int64_t)blockIdx.x * m.rpp + rr; p < positions; p += (int64_t)gridDim.x * m.rpp) {
      float x[8], g[8];
      load8<T>(logits + p * ld + j0, x);
      int a = a0, k = k0;
      int t = a < na ? tgt[p * na + a] : -2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        g[e] = 0.f;
        if (j0 + e < nch && t != -2) {
          const bool pos = (t == k);
          const float u = pos ? -x[e] : x[e];
          const float ex = __builtin_amdgcn_exp2f(-1.44269504f * fabsf(u));
          const float inv = __builtin_amdgcn_rcpf(1.f + ex);
          const float sg = u >= 0.f ? inv : ex * inv;
          float sp = fmaxf(u, 0.f) + 0.69314718f * __builtin_amdgcn_logf(1.f + ex);
          if (LS) sp = fmaf(-half_ls, u, sp);
          const float af = pos ? alpha : 1.f - alpha;
          const float mod = G15 ? sg * __builtin_amdgcn_sqrtf(sg) : __powf(sg, gamma);
          loss_acc = fmaf(af * mod, sp * inv_norm, loss_acc);
          const float dldu = af * mod * fmaf(gamma * (1.f - sg), sp, LS ? sg - half_ls : sg) * inv_norm;
          g[e] = pos ? -dldu : dldu;
        }
        db[e] += g[e];
        if (++k == nc) {
          k = 0;
          ++a;
          t = a < na ? tgt[p * na + a] : -2;
        }
      }
      store8<T>(dlogits + p * ld + j0, g);
    }
  } else if (ok) {
    // One row per step, the next row's logits and targets requested before this row is worked on (r03: the first
    // version loaded, computed and stored one row at a time and fetched the second anchor's target in the middle of the
    // element loop -- a dependent global load under a branch; 85 % "VALU busy" was mostly that wait).  An 8-element
    // chunk spans at most two anchors (num_classes >= 8): both targets are loaded up front, the element loop is
    // branch-free.
    const int64_t step = (int64_t)gridDim.x * m.rpp;
    int64_t p = (int64_t)blockIdx.x * m.rpp + rr;
    const bool has_a1 = a0 + 1 < na;
    float x[8], xn[8];
    int t0 = -2, t1 = -2, t0n = -2, t1n = -2;
    if (p < positions) {
      load8<T>(logits + p * ld + j0, x);
      t0 = a0 < na ? tgt[p * na + a0] : -2;
      t1 = has_a1 ? tgt[p * na + a0 + 1] : -2;
    }
    while (p < positions) {
      const int64_t pn = p + step;
      if (pn < positions) {
        load8<T>(logits + pn * ld + j0, xn);
        t0n = a0 < na ? tgt[pn * na + a0] : -2;
        t1n = has_a1 ? tgt[pn * na + a0 + 1] : -2;
      }
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = k0 + e;
        const bool cross = idx >= nc;
        const int k = cross ? idx - nc : idx;
        const int t = cross ? t1 : t0;
        const bool pos = (t == k);
        const bool valid = (j0 + e < nch) && t != -2;
        const float xe = valid ? x[e] : 0.f;                 // padding columns / ignored anchors may hold anything
        const float u = pos ? -xe : xe;
        // hardware transcendentals (1 ulp: v_exp_f32, v_rcp_f32, v_log_f32, v_sqrt_f32); the IEEE-rounded
        // library forms (__frcp_rn, __fsqrt_rn, __logf) expanded to ~35 extra VALU instructions per logit and
        // made this kernel 100 % VALU-bound.  1 + ex is in [1, 2]: no denormal handling is needed.
        const float ex = __builtin_amdgcn_exp2f(-1.44269504f * fabsf(u));
        const float inv = __builtin_amdgcn_rcpf(1.f + ex);
        const float sg = u >= 0.f ? inv : ex * inv;          // sigmoid(u) = 1 - p_t
        float sp = fmaxf(u, 0.f) + 0.69314718f * __builtin_amdgcn_logf(1.f + ex);   // softplus(u) = cross entropy
        if (LS) sp = fmaf(-half_ls, u, sp);                    // ... against the smoothed label
        const float af = pos ? alpha : 1.f - alpha;
        const float mod = G15 ? sg * __builtin_amdgcn_sqrtf(sg) : __powf(sg, gamma);
        const float wgt = valid ? af * mod * inv_norm : 0.f;
        loss_acc = fmaf(wgt, sp, loss_acc);
        // d/du [sg^gamma * softplus(u)] = sg^gamma * (gamma*(1-sg)*sp + sg)
        const float dldu = wgt * fmaf(gamma * (1.f - sg), sp, LS ? sg - half_ls : sg);
        g[e] = pos ? -dldu : dldu;
        db[e] += g[e];
      }
      store8<T>(dlogits + p * ld + j0, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = xn[e];
      t0 = t0n; t1 = t1n;
      p = pn;
    }
  }
  extern __shared__ float red[];  // [THREADS * 8]
  loss_tail(loss_acc, db, ok, nch, m, red, part, &sums[0], dbias);
}

template <typename T, bool G15>
__global__ __launch_bounds__(THREADS) void k_focal(const T* __restrict__ logits, int ld,
                                                  const int32_t* __restrict__ tgt, int64_t positions,
                                                  int na, int nc, float alpha, float gamma, float inv_norm_h,
                                                  const float* __restrict__ norm_scale,
                                                  T* __restrict__ dlogits, float* dbias, float* sums, float* part, RowMap m) {
  focal_body<T, G15, false>(logits, ld, tgt, positions, na, nc, alpha, gamma, inv_norm_h, norm_scale, dlogits, dbias, sums, part, m, 0.f);
}
template <typename T, bool G15>
__global__ __launch_bounds__(THREADS) void k_focal_ls(const T* __restrict__ logits, int ld,
                                                     const int32_t* __restrict__ tgt, int64_t positions,
                                                     int na, int nc, float alpha, float gamma, float inv_norm_h,
                                                     const float* __restrict__ norm_scale,
                                                     T* __restrict__ dlogits, float* dbias, float* sums, float* part, RowMap m,
                                                     float half_ls) {
  focal_body<T, G15, true>(logits, ld, tgt, positions, na, nc, alpha, gamma, inv_norm_h, norm_scale, dlogits, dbias, sums, part, m, half_ls);
}

template <typename T>
__global__ __launch_bounds__(THREADS) void k_box(const T* __restrict__ out, int ld,
                                                const float* __restrict__ tgt, int64_t positions, int nch,
                                                float delta, float inv_norm_h, float grad_scale,
                                                const float* __restrict__ norm_scale,
                                                T* __restrict__ dbox, float* dbias, float* sums, float* part, RowMap m) {
  const float inv_norm = norm_scale ? inv_norm_h * norm_scale[0] : inv_norm_h;
  const int tid = threadIdx.x;
  const int cv = tid % m.tpr, rr = tid / m.tpr;
  const int j0 = cv * 8;
  const bool ok = j0 < ld;
  float loss_acc = 0.f, db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) db[e] = 0.f;
  if (ok) {
    for (int64_t p = (int64_t)blockIdx.x * m.rpp + rr; p < positions; p += (int64_t)gridDim.x * m.rpp) {
      float x[8], g[8];
      load8<T>(out + p * ld + j0, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = j0 + e;
        g[e] = 0.f;
        if (j < nch) {
          const float t = tgt[p * nch + j];
          if (t != 0.f) {
            const float err = x[e] - t;
            const float ae = fabsf(err);
            const bool quad = ae <= delta;
            loss_acc += (quad ? 0.5f * err * err : delta * ae - 0.5f * delta * delta) * inv_norm;
            g[e] = (quad ? err : (err > 0.f ? delta : -delta)) * inv_norm * grad_scale;
          }
        }
        db[e] += g[e];
      }
      store8<T>(dbox + p * ld + j0, g);
    }
  }
  extern __shared__ float red[];  // [THREADS * 8]
  loss_tail(loss_acc, db, ok, nch, m, red, part, &sums[1], dbias);
}

// ----------------------------------------------------------------------------------- optimizer
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < THREADS / 64; ++i) t += sh[i];
  return t;
}

// Every tensor segment is cut into up to OPT_SPLIT slices of >= 1024 elements, one workgroup per
// (segment, slice): the few large kernels (>100 k elements) no longer serialise on one workgroup, the
// many small tensors still cost one workgroup each, and every reduction keeps a fixed order.
constexpr int OPT_SPLIT = EDET_OPT_SPLIT;

__device__ __forceinline__ bool slice_range(const int64_t* seg_off, int s, int j, int64_t& b, int64_t& e) {
  const int64_t sb = seg_off[s], se = seg_off[s + 1];
  int64_t chunk = ((se - sb + OPT_SPLIT - 1) / OPT_SPLIT + 3) / 4 * 4;
  if (chunk < 1024) chunk = 1024;
  b = sb + (int64_t)j * chunk;
  e = b + chunk < se ? b + chunk : se;
  return b < e;
}

__global__ __launch_bounds__(THREADS) void k_l2_norms(float* grads, const float* params,
                                                     const int64_t* seg_off, const int32_t* seg_flags,
                                                     float wd, float* seg_sqnorm, int nseg) {
  __shared__ float sh[THREADS / 64];
  const int s = blockIdx.x, j = blockIdx.y;
  int64_t b, e;
  const bool any = slice_range(seg_off, s, j, b, e);
  const bool frozen = (seg_flags[s] & EDET_SEG_FROZEN) != 0;
  const bool reg = (seg_flags[s] & EDET_SEG_L2) != 0 && !frozen;
  float gsq = 0.f, wsq = 0.f;
  if (any && frozen) {
    // a frozen variable (config.var_freeze_expr) has no gradient: zeroed here, no share in the norms, skipped by the update
    for (int64_t i = b + threadIdx.x; i < e; i += THREADS) grads[i] = 0.f;
  } else if (any) {
    if ((b & 3) == 0) {
      const int64_t nv = (e - b) >> 2;
      float4* g4 = reinterpret_cast<float4*>(grads + b);
      const float4* w4 = reinterpret_cast<const float4*>(params + b);
      for (int64_t i = threadIdx.x; i < nv; i += THREADS) {
        float4 g = g4[i];
        if (reg) {
          const float4 w = w4[i];
          g.x = fmaf(wd, w.x, g.x); g.y = fmaf(wd, w.y, g.y); g.z = fmaf(wd, w.z, g.z); g.w = fmaf(wd, w.w, g.w);
          g4[i] = g;
          wsq += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
        }
        gsq += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
      }
      b += nv << 2;
    }
    for (int64_t i = b + threadIdx.x; i < e; i += THREADS) {
      float g = grads[i];
      if (reg) {
        const float w = params[i];
        g = fmaf(wd, w, g);
        grads[i] = g;
        wsq = fmaf(w, w, wsq);
      }
      gsq = fmaf(g, g, gsq);
    }
  }
  const float tg = block_sum(gsq, sh);
  const float tw = block_sum(wsq, sh);
  if (threadIdx.x == 0) {
    seg_sqnorm[(size_t)s * OPT_SPLIT + j] = tg;
    // this slice's share of the L2 loss, summed in a fixed order by k_clip_factors (r04: no atomics)
    seg_sqnorm[((size_t)nseg + s) * OPT_SPLIT + j] = (any && reg) ? 0.5f * wd * tw : 0.f;
  }
}

// tf.clip_by_norm per tensor, then tf.clip_by_global_norm over the clipped tensors
__global__ __launch_bounds__(THREADS) void k_clip_factors(const float* seg_sqnorm, int nseg, float clip,
                                                         float* seg_factor, float* gnorm_out, float* l2_sum) {
  __shared__ float sh[THREADS / 64];
  float acc = 0.f, l2 = 0.f;
  for (int s = threadIdx.x; s < nseg; s += THREADS) {
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < OPT_SPLIT; ++j) {
      sq += seg_sqnorm[(size_t)s * OPT_SPLIT + j];
      l2 += seg_sqnorm[((size_t)nseg + s) * OPT_SPLIT + j];
    }
    const float nrm = sqrtf(sq);
    float f = 1.f;
    if (clip > 0.f) f = clip / fmaxf(nrm, clip);
    seg_factor[s] = f;
    const float cn = nrm * f;
    acc = fmaf(cn, cn, acc);
  }
  const float tot = block_sum(acc, sh);
  const float l2tot = block_sum(l2, sh);
  if (threadIdx.x == 0 && l2_sum) l2_sum[0] += l2tot;
  const float gn = sqrtf(tot);
  float f2 = 1.f;
  if (clip > 0.f) f2 = clip / fmaxf(gn, clip);
  __syncthreads();
  for (int s = threadIdx.x; s < nseg; s += THREADS) seg_factor[s] *= f2;
  if (threadIdx.x == 0 && gnorm_out) gnorm_out[0] = gn * f2;
}

__global__ __launch_bounds__(THREADS) void k_scale(float* grads, const int64_t* seg_off,
                                                  const float* seg_factor) {
  const int s = blockIdx.x;
  int64_t b, e;
  if (!slice_range(seg_off, s, blockIdx.y, b, e)) return;
  const float f = seg_factor[s];
  if ((b & 3) == 0) {
    const int64_t nv = (e - b) >> 2;
    float4* g4 = reinterpret_cast<float4*>(grads + b);
    for (int64_t i = threadIdx.x; i < nv; i += THREADS) {
      float4 g = g4[i];
      g.x *= f; g.y *= f; g.z *= f; g.w *= f;
      g4[i] = g;
    }
    b += nv << 2;
  }
  for (int64_t i = b + threadIdx.x; i < e; i += THREADS) grads[i] *= f;
}

// Keras SGD: v = m*v - lr*g ; w += v.  TFA MovingAverage: ema -= (1-decay)*(ema - w)
__device__ __forceinline__ void sgd1(float g, float& v, float& w, float& em, float lr, float momentum,
                                     float decay, bool has_ema) {
  v = momentum * v - lr * g;
  w += v;
  if (has_ema) em -= (1.f - decay) * (em - w);
}

__global__ __launch_bounds__(THREADS) void k_sgd_ema(float* params, float* grads, float* vel, float* ema,
                                                    const int64_t* seg_off, const float* seg_factor,
                                                    const int32_t* seg_flags, const float* hyper, float momentum) {
  const int s = blockIdx.x;
  int64_t b, e;
  if (!slice_range(seg_off, s, blockIdx.y, b, e)) return;
  // frozen variables are not in the optimizer's variable list (tf2/train_lib.py:478-491,683): value, momentum slot and
  // EMA shadow stay exactly as they are
  if (seg_flags && (seg_flags[s] & EDET_SEG_FROZEN)) return;
  const float f = seg_factor ? seg_factor[s] : 1.f;
  const float lr = hyper[0], decay = hyper[1];
  const bool has_ema = ema != nullptr;
  if ((b & 3) == 0) {
    const int64_t nv = (e - b) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(grads + b);
    float4* v4 = reinterpret_cast<float4*>(vel + b);
    float4* w4 = reinterpret_cast<float4*>(params + b);
    float4* e4 = has_ema ? reinterpret_cast<float4*>(ema + b) : nullptr;
    for (int64_t i = threadIdx.x; i < nv; i += THREADS) {
      const float4 g = g4[i];
      float4 v = v4[i], w = w4[i], em = has_ema ? e4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      sgd1(g.x * f, v.x, w.x, em.x, lr, momentum, decay, has_ema);
      sgd1(g.y * f, v.y, w.y, em.y, lr, momentum, decay, has_ema);
      sgd1(g.z * f, v.z, w.z, em.z, lr, momentum, decay, has_ema);
      sgd1(g.w * f, v.w, w.w, em.w, lr, momentum, decay, has_ema);
      v4[i] = v;
      w4[i] = w;
      if (has_ema) e4[i] = em;
    }
    b += nv << 2;
  }
  for (int64_t i = b + threadIdx.x; i < e; i += THREADS) {
    float v = vel[i], w = params[i], em = has_ema ? ema[i] : 0.f;
    sgd1(grads[i] * f, v, w, em, lr, momentum, decay, has_ema);
    vel[i] = v;
    params[i] = w;
    if (has_ema) ema[i] = em;
  }
}

}  // namespace

extern "C" int edet_focal_loss_smooth(const void* logits, int ld, const int32_t* cls_targets,
                                      int64_t positions, int num_anchors, int num_classes,
                                      float alpha, float gamma, float label_smoothing, float inv_normalizer,
                                      const float* norm_scale_dev,
                                      void* dlogits, float* dbias, float* sums, void* workspace, size_t workspace_bytes,
                                      int dtype, void* stream) {
  EDET_CHECK(logits && cls_targets && dlogits && sums, "edet_focal_loss: null pointer");
  EDET_CHECK(ld % 8 == 0 && ld >= num_anchors * num_classes && ld <= 2048, "edet_focal_loss: bad ld %d", ld);
  EDET_CHECK(num_classes >= 1, "edet_focal_loss: num_classes must be positive");
  EDET_CHECK(label_smoothing >= 0.f && label_smoothing <= 1.f, "edet_focal_loss: label_smoothing %g outside [0, 1]", (double)label_smoothing);
  const RowMap m = row_map_ld(ld);
  int64_t g = (positions + m.rpp - 1) / m.rpp;
  g = (g + 3) / 4;
  const size_t lds = (size_t)THREADS * 8 * sizeof(float);
  const bool g15 = gamma == 1.5f;
  const bool ls = label_smoothing != 0.f;
  const float half_ls = 0.5f * label_smoothing;
  {   // one round of the workgroups the chip holds at once (the rows are strided over the grid)
    const void* fn = dtype == EDET_BF16 ? (g15 ? reinterpret_cast<const void*>(&k_focal<bf16_t, true>) : reinterpret_cast<const void*>(&k_focal<bf16_t, false>))
                                        : (g15 ? reinterpret_cast<const void*>(&k_focal<float, true>) : reinterpret_cast<const void*>(&k_focal<float, false>));
    const int slots = edet_resident_wgs(fn, THREADS, lds);
    const int64_t cap = slots > 0 ? slots : 4096;
    if (g > cap) g = cap;
  }
  if (g < 1) g = 1;
  // ordered partial rows [g][1 + nch] when the workspace holds them (else: one workgroup)
  const int nch_ = num_anchors * num_classes;
  float* part = (workspace && workspace_bytes >= (size_t)g * (1 + nch_) * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
  if (!part) g = 1;
#define FOCAL_LAUNCH(T, G)                                                                            \
  edet_launch(k_focal<T, G>, dim3((int)g), dim3(THREADS), lds, to_stream(stream), (const T*)logits, ld, cls_targets, positions, \
      num_anchors, num_classes, alpha, gamma, inv_normalizer, norm_scale_dev, (T*)dlogits, dbias, sums, part, m)
#define FOCAL_LAUNCH_LS(T, G)                                                                         \
  edet_launch(k_focal_ls<T, G>, dim3((int)g), dim3(THREADS), lds, to_stream(stream), (const T*)logits, ld, cls_targets, positions, \
      num_anchors, num_classes, alpha, gamma, inv_normalizer, norm_scale_dev, (T*)dlogits, dbias, sums, part, m, half_ls)
#define FOCAL_LAUNCH_T(T)                                                  \
  do {                                                                     \
    if (g15) { if (ls) FOCAL_LAUNCH_LS(T, true); else FOCAL_LAUNCH(T, true); }     \
    else { if (ls) FOCAL_LAUNCH_LS(T, false); else FOCAL_LAUNCH(T, false); }       \
  } while (0)
  if (dtype == EDET_BF16) FOCAL_LAUNCH_T(bf16_t);
  else if (dtype == EDET_F32) FOCAL_LAUNCH_T(float);
#undef FOCAL_LAUNCH_T
#undef FOCAL_LAUNCH_LS
#undef FOCAL_LAUNCH
  else EDET_CHECK(false, "edet_focal_loss: bad dtype %d", dtype);
  if (part && edet_reduce_partials2(part, (int)g, 1 + nch_, dbias, nch_, &sums[0], to_stream(stream)) != 0) return -2;
  EDET_LAUNCH_CHECK("edet_focal_loss");
  return 0;
}

extern "C" int edet_focal_loss(const void* logits, int ld, const int32_t* cls_targets,
                               int64_t positions, int num_anchors, int num_classes,
                               float alpha, float gamma, float inv_normalizer,
                               const float* norm_scale_dev,
                               void* dlogits, float* dbias, float* sums, void* workspace, size_t workspace_bytes,
                               int dtype, void* stream) {
  return edet_focal_loss_smooth(logits, ld, cls_targets, positions, num_anchors, num_classes, alpha, gamma, 0.f,
                                inv_normalizer, norm_scale_dev, dlogits, dbias, sums, workspace, workspace_bytes, dtype,
                                stream);
}

extern "C" int edet_box_loss(const void* box_out, int ld, const float* box_targets,
                             int64_t positions, int nch, float delta, float inv_normalizer,
                             float grad_scale, const float* norm_scale_dev, void* dbox, float* dbias,
                             float* sums, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  EDET_CHECK(box_out && box_targets && dbox && sums, "edet_box_loss: null pointer");
  EDET_CHECK(ld % 8 == 0 && ld >= nch && ld <= 2048, "edet_box_loss: bad ld %d", ld);
  const RowMap m = row_map_ld(ld);
  int64_t g = (positions + m.rpp - 1) / m.rpp;
  g = (g + 3) / 4;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  const size_t lds = (size_t)THREADS * 8 * sizeof(float);
  float* part = (workspace && workspace_bytes >= (size_t)g * (1 + nch) * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
  if (!part) g = 1;
  if (dtype == EDET_BF16)
    edet_launch(k_box<bf16_t>, dim3((int)g), dim3(THREADS), lds, to_stream(stream), (const bf16_t*)box_out, ld, box_targets, positions, nch, delta, inv_normalizer, grad_scale, norm_scale_dev, (bf16_t*)dbox, dbias, sums, part, m);
  else if (dtype == EDET_F32)
    edet_launch(k_box<float>, dim3((int)g), dim3(THREADS), lds, to_stream(stream), (const float*)box_out, ld, box_targets, positions, nch, delta, inv_normalizer, grad_scale, norm_scale_dev, (float*)dbox, dbias, sums, part, m);
  else EDET_CHECK(false, "edet_box_loss: bad dtype %d", dtype);
  if (part && edet_reduce_partials2(part, (int)g, 1 + nch, dbias, nch, &sums[1], to_stream(stream)) != 0) return -2;
  EDET_LAUNCH_CHECK("edet_box_loss");
  return 0;
}

extern "C" int edet_opt_l2_norms(float* grads, const float* params, const int64_t* seg_offsets,
                                 const int32_t* seg_flags, int nseg, float weight_decay,
                                 float* seg_sqnorm, void* stream) {
  EDET_CHECK(grads && params && seg_offsets && seg_flags && seg_sqnorm && nseg > 0, "edet_opt_l2_norms: bad arguments");
  edet_launch(k_l2_norms, dim3(nseg, OPT_SPLIT), dim3(THREADS), 0, to_stream(stream), grads, params, seg_offsets, seg_flags, weight_decay, seg_sqnorm, nseg);
  EDET_LAUNCH_CHECK("edet_opt_l2_norms");
  return 0;
}

extern "C" int edet_opt_clip_factors(const float* seg_sqnorm, int nseg, float clip_norm,
                                     float* seg_factor, float* global_norm_out, float* l2_sum, void* stream) {
  EDET_CHECK(seg_sqnorm && seg_factor && nseg > 0, "edet_opt_clip_factors: bad arguments");
  edet_launch(k_clip_factors, dim3(1), dim3(THREADS), 0, to_stream(stream), seg_sqnorm, nseg, clip_norm, seg_factor, global_norm_out, l2_sum);
  EDET_LAUNCH_CHECK("edet_opt_clip_factors");
  return 0;
}

extern "C" int edet_opt_scale(float* grads, const int64_t* seg_offsets, const float* seg_factor,
                              int nseg, void* stream) {
  EDET_CHECK(grads && seg_offsets && seg_factor && nseg > 0, "edet_opt_scale: bad arguments");
  edet_launch(k_scale, dim3(nseg, OPT_SPLIT), dim3(THREADS), 0, to_stream(stream), grads, seg_offsets, seg_factor);
  EDET_LAUNCH_CHECK("edet_opt_scale");
  return 0;
}

namespace {
// tf.keras.optimizers.Adam (ResourceApplyAdam): m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); w -= alpha m / (sqrt(v) + eps)
// with alpha = lr sqrt(1 - b2^t) / (1 - b1^t) formed by the host for this step (hyper[0]); TFA MovingAverage on top
__device__ __forceinline__ void adam1(float g, float& m, float& u, float& w, float& em, float alpha, float b1, float b2,
                                      float eps, float decay, bool has_ema) {
  m += (g - m) * (1.f - b1);
  u += (g * g - u) * (1.f - b2);
  w -= (m * alpha) / (sqrtf(u) + eps);
  if (has_ema) em -= (1.f - decay) * (em - w);
}

__global__ __launch_bounds__(THREADS) void k_adam_ema(float* params, const float* grads, float* m1, float* m2, float* ema,
                                                     const int64_t* seg_off, const float* seg_factor,
                                                     const int32_t* seg_flags, const float* hyper, float b1, float b2, float eps) {
  const int s = blockIdx.x;
  int64_t b, e;
  if (!slice_range(seg_off, s, blockIdx.y, b, e)) return;
  if (seg_flags && (seg_flags[s] & EDET_SEG_FROZEN)) return;      // not in the optimizer's variable list
  const float f = seg_factor ? seg_factor[s] : 1.f;
  const float alpha = hyper[0], decay = hyper[1];
  const bool has_ema = ema != nullptr;
  for (int64_t i = b + threadIdx.x; i < e; i += THREADS) {
    float m = m1[i], u = m2[i], w = params[i], em = has_ema ? ema[i] : 0.f;
    adam1(grads[i] * f, m, u, w, em, alpha, b1, b2, eps, decay, has_ema);
    m1[i] = m;
    m2[i] = u;
    params[i] = w;
    if (has_ema) ema[i] = em;
  }
}
}  // namespace

extern "C" int edet_opt_adam_ema(float* params, const float* grads, float* m, float* v, float* ema,
                                 const int64_t* seg_offsets, const float* seg_factor, const int32_t* seg_flags, int nseg,
                                 const float* hyper_dev, float beta1, float beta2, float epsilon, void* stream) {
  EDET_CHECK(params && grads && m && v && seg_offsets && hyper_dev && nseg > 0, "edet_opt_adam_ema: bad arguments");
  edet_launch(k_adam_ema, dim3(nseg, OPT_SPLIT), dim3(THREADS), 0, to_stream(stream), params, grads, m, v, ema, seg_offsets,
              seg_factor, seg_flags, hyper_dev, beta1, beta2, epsilon);
  EDET_LAUNCH_CHECK("edet_opt_adam_ema");
  return 0;
}

extern "C" int edet_opt_sgd_ema(float* params, float* grads, float* velocity, float* ema,
                                const int64_t* seg_offsets, const float* seg_factor, const int32_t* seg_flags, int nseg,
                                const float* hyper_dev, float momentum, void* stream) {
  EDET_CHECK(params && grads && velocity && seg_offsets && hyper_dev && nseg > 0, "edet_opt_sgd_ema: bad arguments");
  edet_launch(k_sgd_ema, dim3(nseg, OPT_SPLIT), dim3(THREADS), 0, to_stream(stream), params, grads, velocity, ema, seg_offsets, seg_factor, seg_flags, hyper_dev, momentum);
  EDET_LAUNCH_CHECK("edet_opt_sgd_ema");
  return 0;
}

// ---- step plumbing that used to be torch calls (round 6: every launch of a step goes through this ABI, so that a step
// can be recorded and replayed by a host without a Python interpreter: net_runtime.cpp) ---------------------------------
namespace {
__global__ __launch_bounds__(THREADS) void k_axpy_clear(float* __restrict__ dst, float* __restrict__ src, int64_t n, int clear) {
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) {
    dst[i] += src[i];
    if (clear) src[i] = 0.f;
  }
}
// inv_out[0] = 1 / (sum_i mean_num_positives[i] + 1): one wave, lanes in a fixed order (the counts are integers: exact)
__global__ __launch_bounds__(64) void k_loss_normalizer(const float* __restrict__ mnp, int n, float* __restrict__ inv_out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += mnp[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) inv_out[0] = 1.0f / (s + 1.0f);
}
}  // namespace

namespace {
__global__ __launch_bounds__(THREADS) void k_widen_bf16(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride)
    dst[i] = __uint_as_float((uint32_t)src[i] << 16);
}
}  // namespace

extern "C" int edet_cast_to_f32(const void* src, float* dst, int64_t count, int src_dtype, void* stream) {
  EDET_CHECK(src && dst, "edet_cast_to_f32: null pointer");
  if (count <= 0) return 0;
  if (src_dtype == EDET_F32) {
    const hipError_t e = hipMemcpyAsync(dst, src, (size_t)count * 4, hipMemcpyDeviceToDevice, to_stream(stream));
    EDET_CHECK(e == hipSuccess, "edet_cast_to_f32: hipMemcpyAsync: %s", hipGetErrorString(e));
    return 0;
  }
  EDET_CHECK(src_dtype == EDET_BF16, "edet_cast_to_f32: bad dtype %d", src_dtype);
  int64_t grid = (count + THREADS - 1) / THREADS;
  if (grid > 4096) grid = 4096;
  edet_launch(k_widen_bf16, dim3((unsigned)grid), dim3(THREADS), 0, to_stream(stream), (const bf16_t*)src, dst, count);
  EDET_LAUNCH_CHECK("edet_cast_to_f32");
  return 0;
}

namespace {
// dst[r][0 .. cw) = src[r][0 .. cw) in 4-byte words: a [rows][ld] tensor without its padding columns
__global__ __launch_bounds__(THREADS) void k_compact_rows(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                         int64_t rows, int cw, int ldw) {
  const int64_t total = rows * cw, stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cw;
    dst[i] = src[r * ldw + (i - r * cw)];
  }
}
}  // namespace

extern "C" int edet_compact_rows(const void* src, int64_t rows, int c, int ld, void* dst, int elem_bytes, void* stream) {
  EDET_CHECK(src && dst && rows >= 0 && c > 0 && ld >= c, "edet_compact_rows: bad arguments");
  EDET_CHECK((elem_bytes == 2 || elem_bytes == 4) && (c * elem_bytes) % 4 == 0 && (ld * elem_bytes) % 4 == 0,
             "edet_compact_rows: rows must be whole 4-byte words (elem_bytes %d, c %d, ld %d)", elem_bytes, c, ld);
  if (rows == 0) return 0;
  const int cw = c * elem_bytes / 4, ldw = ld * elem_bytes / 4;
  int64_t grid = (rows * cw + THREADS - 1) / THREADS;
  if (grid > 4096) grid = 4096;
  edet_launch(k_compact_rows, dim3((unsigned)grid), dim3(THREADS), 0, to_stream(stream), (const uint32_t*)src, (uint32_t*)dst,
              rows, cw, ldw);
  EDET_LAUNCH_CHECK("edet_compact_rows");
  return 0;
}

namespace {
// A kernel, not hipMemsetAsync: a memset NODE of a captured graph costs a ~50 us bubble in front of it on this runtime
// (r06zz timeline: two fillBufferAligned nodes and the kernel behind them, 0.2 ms of idle queue per step)
__global__ __launch_bounds__(THREADS) void k_zero(uint4* __restrict__ dst16, size_t n16, unsigned char* __restrict__ tail, int ntail) {
  const size_t stride = (size_t)gridDim.x * THREADS;
  for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n16; i += stride) dst16[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
}  // namespace

extern "C" int edet_zero(void* dst, size_t bytes, void* stream) {
  EDET_CHECK(dst || bytes == 0, "edet_zero: null pointer");
  if (bytes == 0) return 0;
  // head up to the first 16-byte boundary and tail behind the last one: byte stores of block 0 (at most 15 + 15)
  unsigned char* p = reinterpret_cast<unsigned char*>(dst);
  const size_t mis = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
  if (mis >= bytes || mis != 0) {
    // unaligned start (not what the engine passes: its buffers are 256-byte aligned): the memset path
    const hipError_t e = hipMemsetAsync(dst, 0, bytes, to_stream(stream));
    EDET_CHECK(e == hipSuccess, "edet_zero: hipMemsetAsync: %s", hipGetErrorString(e));
    return 0;
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes - n16 * 16);
  size_t grid = (n16 + THREADS - 1) / THREADS;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  edet_launch(k_zero, dim3((unsigned)grid), dim3(THREADS), 0, to_stream(stream), reinterpret_cast<uint4*>(p), n16, p + n16 * 16, ntail);
  EDET_LAUNCH_CHECK("edet_zero");
  return 0;
}

extern "C" int edet_axpy_clear(float* dst, float* src, int64_t n, int clear_src, void* stream) {
  EDET_CHECK(dst && src && n >= 0, "edet_axpy_clear: bad arguments");
  if (n == 0) return 0;
  int64_t grid = (n + THREADS - 1) / THREADS;
  if (grid > 2048) grid = 2048;
  edet_launch(k_axpy_clear, dim3((unsigned)grid), dim3(THREADS), 0, to_stream(stream), dst, src, n, clear_src);
  EDET_LAUNCH_CHECK("edet_axpy_clear");
  return 0;
}

extern "C" int edet_loss_normalizer(const float* mean_num_positives, int n, float* inv_out, void* stream) {
  EDET_CHECK(mean_num_positives && inv_out && n > 0, "edet_loss_normalizer: bad arguments");
  edet_launch(k_loss_normalizer, dim3(1), dim3(64), 0, to_stream(stream), mean_num_positives, n, inv_out);
  EDET_LAUNCH_CHECK("edet_loss_normalizer");
  return 0;
}
