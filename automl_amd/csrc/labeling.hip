// Anchor labelling on device.
//
//
//   edet_label_anchors   tf2/anchors.py AnchorLabeler.label_anchors :215-250 for a batch: IoU of every groundtruth
//                        box with every anchor (object_detection/region_similarity_calculator.py:42-88), ArgMaxMatcher
//                        with matched = unmatched threshold and force_match_for_each_row (argmax_matcher.py:101-184),
//                        class targets (class - 1, -1 background), Faster-RCNN box encoding of the matched box
//                        (faster_rcnn_box_coder.py:59-89), positives per image.
//
// Three launches: (1) one workgroup per (groundtruth box, image): the box's best anchor (first maximum) -> atomicMin of
// the row index into force[b][anchor]; (2) one thread per anchor: loop over the image's boxes held in LDS, first
// maximum, threshold, force-match override, targets written straight into the per-level label tensors; (3) nothing
// else: the positive counts are block-reduced in (2) and added with one atomic per workgroup.
#include <math.h>

#include "common.h"

// The arithmetic below restates float32 numpy / TensorFlow expressions operation by operation (argmax ties and
// 1e-6 parities depend on it): this file is compiled with -ffp-contract=off (automl_amd/build.py) -- hipcc
// contracts a*b+c into an FMA by default, and __fmul_rn / __fadd_rn are plain operators in HIP, not the
// contraction barriers they are in CUDA.

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int LB_THREADS = 256;
constexpr int MAX_GT = 256;

struct LabelArgs {
  const float* anchors;      // [N][4]
  const float* gt_boxes;     // [B][max_gt][4]
  const int* gt_labels;      // [B][max_gt]
  const int* gt_count;       // [B]
  int batch, max_gt, N;
  float threshold;
  int nlevels;
  int aoff[MAX_LEVELS + 1];
  int lanch[MAX_LEVELS];
  int* cls_out[MAX_LEVELS];    // [B][lanch]
  float* box_out[MAX_LEVELS];  // [B][lanch][4]
  int* force;                  // [B][N], preset to INT_MAX
  float* num_positives;        // [B], preset to 0
};

// region_similarity_calculator.iou for one pair, float32, op for op
__device__ __forceinline__ float iou_pair(const float4 g, const float4 a) {
  const float ih = fmaxf(0.f, __fsub_rn(fminf(g.z, a.z), fmaxf(g.x, a.x)));
  const float iw = fmaxf(0.f, __fsub_rn(fminf(g.w, a.w), fmaxf(g.y, a.y)));
  const float inter = __fmul_rn(ih, iw);
  if (inter == 0.f) return 0.f;
  const float a1 = __fmul_rn(__fsub_rn(g.z, g.x), __fsub_rn(g.w, g.y));
  const float a2 = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
  return inter / __fsub_rn(__fadd_rn(a1, a2), inter);
}

__global__ __launch_bounds__(LB_THREADS) void k_label_force(const LabelArgs a) {
  __shared__ float rs[LB_THREADS / 64];
  __shared__ int rp[LB_THREADS / 64];
  const int m = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (m >= min(max(a.gt_count[b], 0), min(a.max_gt, MAX_GT))) return;    // a count beyond the padded rows is clamped
  const float4 g = *reinterpret_cast<const float4*>(a.gt_boxes + ((size_t)b * a.max_gt + m) * 4);
  float best = -1.f;
  int pos = 0x7fffffff;
  for (int n = tid; n < a.N; n += LB_THREADS) {
    const float v = iou_pair(g, *reinterpret_cast<const float4*>(a.anchors + (size_t)n * 4));
    if (v > best) { best = v; pos = n; }          // ascending n per thread: the first maximum of the thread
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int op = __shfl_xor(pos, off);
    if (ov > best || (ov == best && op < pos)) { best = ov; pos = op; }
  }
  if (lane == 0) { rs[wave] = best; rp[wave] = pos; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < LB_THREADS / 64; ++w)
      if (rs[w] > best || (rs[w] == best && rp[w] < pos)) { best = rs[w]; pos = rp[w]; }
    atomicMin(&a.force[(size_t)b * a.N + pos], m);      // several boxes on one anchor: the lowest row wins
  }
}

__global__ __launch_bounds__(LB_THREADS) void k_label_assign(const LabelArgs a) {
  __shared__ float4 gbox[MAX_GT];
  __shared__ int glab[MAX_GT];
  __shared__ int wcount[LB_THREADS / 64];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = min(max(a.gt_count[b], 0), min(a.max_gt, MAX_GT));    // device value: never trust it past the padded rows / LDS arrays
  for (int m = tid; m < M; m += LB_THREADS) {
    gbox[m] = *reinterpret_cast<const float4*>(a.gt_boxes + ((size_t)b * a.max_gt + m) * 4);
    glab[m] = a.gt_labels[(size_t)b * a.max_gt + m];
  }
  __syncthreads();
  const int n = blockIdx.x * LB_THREADS + tid;
  int positive = 0;
  if (n < a.N) {
    const float4 an = *reinterpret_cast<const float4*>(a.anchors + (size_t)n * 4);
    float best = -1.f;
    int match = -1;
    for (int m = 0; m < M; ++m) {
      const float v = iou_pair(gbox[m], an);
      if (v > best) { best = v; match = m; }        // tf.argmax over the rows: the first maximum
    }
    if (M == 0 || a.threshold > best) match = -1;     // below the (un)matched threshold -> unmatched
    const int forced = a.force[(size_t)b * a.N + n];
    if (forced != 0x7fffffff) match = forced;
    int l = 0;
    while (l + 1 < a.nlevels && n >= a.aoff[l + 1]) ++l;
    const size_t o = (size_t)b * a.lanch[l] + (n - a.aoff[l]);
    float4 code = make_float4(0.f, 0.f, 0.f, 0.f);
    int cls = -1;
    if (match >= 0) {
      positive = 1;
      cls = glab[match] - 1;
      const float4 g = gbox[match];
      // FasterRcnnBoxCoder._encode with box_list.get_center_coordinates_and_sizes
      const float wa0 = __fsub_rn(an.w, an.y), ha0 = __fsub_rn(an.z, an.x);
      const float yca = __fadd_rn(an.x, __fmul_rn(ha0, 0.5f)), xca = __fadd_rn(an.y, __fmul_rn(wa0, 0.5f));
      const float w0 = __fsub_rn(g.w, g.y), h0 = __fsub_rn(g.z, g.x);
      const float yc = __fadd_rn(g.x, __fmul_rn(h0, 0.5f)), xc = __fadd_rn(g.y, __fmul_rn(w0, 0.5f));
      const float ha = fmaxf(1e-8f, ha0), wa = fmaxf(1e-8f, wa0), h = fmaxf(1e-8f, h0), w = fmaxf(1e-8f, w0);
      code = make_float4(__fsub_rn(yc, yca) / ha, __fsub_rn(xc, xca) / wa, logf(h / ha), logf(w / wa));
    }
    a.cls_out[l][o] = cls;
    *reinterpret_cast<float4*>(a.box_out[l] + o * 4) = code;
  }
  const unsigned long long mask = __ballot(positive);
  if (lane == 0) wcount[wave] = __popcll(mask);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < LB_THREADS / 64; ++w) t += wcount[w];
    if (t) atomicAdd(&a.num_positives[b], (float)t);
  }
}

__global__ void k_label_init(int* force, float* num_positives, int64_t total, int batch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) force[i] = 0x7fffffff;
  if (i < batch) num_positives[i] = 0.f;
}

}  // namespace

extern "C" int edet_label_anchors_workspace_bytes(int batch, int num_anchors, size_t* bytes) {
  EDET_CHECK(batch >= 1 && num_anchors >= 1 && bytes, "edet_label_anchors: bad sizes");
  *bytes = (size_t)batch * num_anchors * 4;
  return 0;
}

extern "C" int edet_label_anchors(const float* anchor_boxes, const int* level_anchors, int nlevels,
                                  const float* gt_boxes, const int* gt_labels, const int* gt_count, int batch,
                                  int max_gt, float match_threshold, void* workspace, size_t workspace_bytes,
                                  int* const* cls_targets, float* const* box_targets, float* num_positives,
                                  void* stream) {
  EDET_CHECK(anchor_boxes && level_anchors && gt_boxes && gt_labels && gt_count && cls_targets && box_targets &&
             num_positives, "edet_label_anchors: null argument");
  EDET_CHECK(nlevels >= 1 && nlevels <= MAX_LEVELS, "edet_label_anchors: %d levels (max %d)", nlevels, MAX_LEVELS);
  EDET_CHECK(max_gt >= 1 && max_gt <= MAX_GT, "edet_label_anchors: max_gt = %d (1..%d)", max_gt, MAX_GT);
  LabelArgs a;
  a.anchors = anchor_boxes; a.gt_boxes = gt_boxes; a.gt_labels = gt_labels; a.gt_count = gt_count;
  a.batch = batch; a.max_gt = max_gt; a.threshold = match_threshold; a.nlevels = nlevels;
  int64_t run = 0;
  for (int l = 0; l < nlevels; ++l) {
    EDET_CHECK(level_anchors[l] > 0 && cls_targets[l] && box_targets[l], "edet_label_anchors: level %d is empty", l);
    a.aoff[l] = (int)run;
    a.lanch[l] = level_anchors[l];
    a.cls_out[l] = cls_targets[l];
    a.box_out[l] = box_targets[l];
    run += level_anchors[l];
  }
  a.aoff[nlevels] = (int)run;
  a.N = (int)run;
  size_t need = 0;
  if (edet_label_anchors_workspace_bytes(batch, a.N, &need)) return -1;
  EDET_CHECK(workspace && workspace_bytes >= need, "edet_label_anchors: workspace %zu < %zu bytes", workspace_bytes, need);
  a.force = reinterpret_cast<int*>(workspace);
  a.num_positives = num_positives;
  hipStream_t st = to_stream(stream);
  const int64_t total = (int64_t)batch * a.N;
  edet_launch(k_label_init, dim3(cdiv(total, 256)), dim3(256), 0, st, a.force, num_positives, total, batch);
  edet_launch(k_label_force, dim3(max_gt, batch), dim3(LB_THREADS), 0, st, a);
  edet_launch(k_label_assign, dim3(cdiv(a.N, LB_THREADS), batch), dim3(LB_THREADS), 0, st, a);
  EDET_LAUNCH_CHECK("edet_label_anchors");
  return 0;
}
