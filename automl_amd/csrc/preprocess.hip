// Image preprocessing on device: inference resize and the training-time DetectionInputProcessor.
//
//   edet_preprocess_infer   efficientdet_keras.EfficientDetModel._preprocessing(mode='infer') :920-951 =
//                           dataloader.InputProcessor.normalize_image :58-64, set_scale_factors_to_output_size
//                           :113-124, resize_and_crop_image :126-139 (tf.image.resize bilinear with half-pixel
//                           centres, tf.image.pad_to_bounding_box) for a batch of equally sized raw images.
//
//   edet_preprocess_train   dataloader.DetectionInputProcessor :144-200 as InputReader.process_example drives it in
//                           training (:321-336): normalize_image, random_horizontal_flip (object_detection/
//                           preprocessor.py:113-199), set_training_random_scale_factors :66-111 (the draws and the scale
//                           arithmetic stay on the host: five integers per image), resize_and_crop_image :126-139,
//                           resize_and_crop_boxes :165-189 with clip_boxes :155-163 and the zero-area filter.
//
// One thread per output pixel: the four taps of each channel are normalised ((v - mean) / stddev, as the reference
// normalises before it resizes) and blended top row, bottom row, then vertically -- the operation order of TF's
// resize_bilinear_op.cc.  Pixels outside the scaled image are zero.  HBM-bound: the raw batch is read once (each
// tap from cache), the output written once.
#include <math.h>

#include "common.h"

// The arithmetic below restates float32 numpy / TensorFlow expressions operation by operation (argmax ties and
// 1e-6 parities depend on it): this file is compiled with -ffp-contract=off (automl_amd/build.py) -- hipcc
// contracts a*b+c into an FMA by default, and __fmul_rn / __fadd_rn are plain operators in HIP, not the
// contraction barriers they are in CUDA.

namespace {

struct PrepArgs {
  const void* raw;      // [B][h][w][3] uint8 or float32
  int raw_is_float;
  int batch, h, w, out_h, out_w, scaled_h, scaled_w;
  float mean[3], stddev[3];
  void* out;            // [B][out_h][out_w][3]
};

__device__ __forceinline__ float raw_at(const PrepArgs& a, size_t idx) {
  return a.raw_is_float ? reinterpret_cast<const float*>(a.raw)[idx]
                        : (float)reinterpret_cast<const unsigned char*>(a.raw)[idx];
}

template <typename T>
__global__ __launch_bounds__(256) void k_preprocess_infer(const PrepArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= a.out_w || y >= a.out_h) return;
  float v[3] = {0.f, 0.f, 0.f};
  if (x < a.scaled_w && y < a.scaled_h) {
    // HalfPixelScaler: in = (out + 0.5) * in_size / out_size - 0.5
    const float sy = __fsub_rn(__fmul_rn(__fadd_rn((float)y, 0.5f), (float)a.h / (float)a.scaled_h), 0.5f);
    const float sx = __fsub_rn(__fmul_rn(__fadd_rn((float)x, 0.5f), (float)a.w / (float)a.scaled_w), 0.5f);
    const float fy = floorf(sy), fx = floorf(sx);
    const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), a.h - 1);
    const int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), a.w - 1);
    const float ly = __fsub_rn(sy, fy), lx = __fsub_rn(sx, fx);
    const size_t img = (size_t)b * a.h * a.w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = __fsub_rn(raw_at(a, ((img + (size_t)y0 * a.w + x0) * 3) + c), a.mean[c]) / a.stddev[c];
      const float tr = __fsub_rn(raw_at(a, ((img + (size_t)y0 * a.w + x1) * 3) + c), a.mean[c]) / a.stddev[c];
      const float bl = __fsub_rn(raw_at(a, ((img + (size_t)y1 * a.w + x0) * 3) + c), a.mean[c]) / a.stddev[c];
      const float br = __fsub_rn(raw_at(a, ((img + (size_t)y1 * a.w + x1) * 3) + c), a.mean[c]) / a.stddev[c];
      const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), lx));
      const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), lx));
      v[c] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ly));
    }
  }
  T* o = reinterpret_cast<T*>(a.out) + (((size_t)b * a.out_h + y) * a.out_w + x) * 3;
  o[0] = from_f<T>(v[0]); o[1] = from_f<T>(v[1]); o[2] = from_f<T>(v[2]);
}

// Training: output pixel (y, x) = pixel (y + offset_y, x + offset_x) of the image resized to [scaled_h, scaled_w]
// (zero beyond it), the raw image mirrored left-right first when flip is set.
template <typename T>
__global__ __launch_bounds__(256) void k_preprocess_train(const PrepArgs a, const edet_prep_image_t* __restrict__ per) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= a.out_w || y >= a.out_h) return;
  const edet_prep_image_t p = per[b];
  const int Y = y + p.offset_y, X = x + p.offset_x;
  float v[3] = {0.f, 0.f, 0.f};
  if (X < p.scaled_w && Y < p.scaled_h) {
    const float sy = __fsub_rn(__fmul_rn(__fadd_rn((float)Y, 0.5f), (float)a.h / (float)p.scaled_h), 0.5f);
    const float sx = __fsub_rn(__fmul_rn(__fadd_rn((float)X, 0.5f), (float)a.w / (float)p.scaled_w), 0.5f);
    const float fy = floorf(sy), fx = floorf(sx);
    const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), a.h - 1);
    int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), a.w - 1);
    if (p.flip) { x0 = a.w - 1 - x0; x1 = a.w - 1 - x1; }       // taps of the mirrored image, read from the raw one
    const float ly = __fsub_rn(sy, fy), lx = __fsub_rn(sx, fx);
    const size_t img = (size_t)b * a.h * a.w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = __fsub_rn(raw_at(a, ((img + (size_t)y0 * a.w + x0) * 3) + c), a.mean[c]) / a.stddev[c];
      const float tr = __fsub_rn(raw_at(a, ((img + (size_t)y0 * a.w + x1) * 3) + c), a.mean[c]) / a.stddev[c];
      const float bl = __fsub_rn(raw_at(a, ((img + (size_t)y1 * a.w + x0) * 3) + c), a.mean[c]) / a.stddev[c];
      const float br = __fsub_rn(raw_at(a, ((img + (size_t)y1 * a.w + x1) * 3) + c), a.mean[c]) / a.stddev[c];
      const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), lx));
      const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), lx));
      v[c] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ly));
    }
  }
  T* o = reinterpret_cast<T*>(a.out) + (((size_t)b * a.out_h + y) * a.out_w + x) * 3;
  o[0] = from_f<T>(v[0]); o[1] = from_f<T>(v[1]); o[2] = from_f<T>(v[2]);
}

// Boxes of one image per workgroup: normalised [ymin, xmin, ymax, xmax] -> [mirrored] -> pixels of the scaled image ->
// minus the crop offset -> clipped to [0, size - 1] -> boxes of zero area dropped, the others kept IN ORDER (tf.where +
// gather_nd); rows past the kept ones are -1 (dataloader.pad_to_fixed_size with -1).
constexpr int PREP_MAX_BOXES = 1024;
__global__ __launch_bounds__(256) void k_preprocess_boxes(const edet_prep_image_t* __restrict__ per,
                                                         const float* __restrict__ boxes_in,
                                                         const float* __restrict__ classes_in,
                                                         const int* __restrict__ counts_in, int max_boxes, int out_h,
                                                         int out_w, float* __restrict__ boxes_out,
                                                         float* __restrict__ classes_out, int* __restrict__ counts_out) {
  __shared__ float4 tb[PREP_MAX_BOXES];
  __shared__ int keep[PREP_MAX_BOXES];
  __shared__ int kept;
  const int b = blockIdx.x, tid = threadIdx.x;
  const edet_prep_image_t p = per[b];
  const int n = min(max(counts_in[b], 0), max_boxes);
  const float sh = (float)p.scaled_h, sw = (float)p.scaled_w, oy = (float)p.offset_y, ox = (float)p.offset_x;
  const float hy = (float)(out_h - 1), hx = (float)(out_w - 1);
  for (int i = tid; i < n; i += blockDim.x) {
    float4 q = *reinterpret_cast<const float4*>(boxes_in + ((size_t)b * max_boxes + i) * 4);
    if (p.flip) {                                     // preprocessor._flip_boxes_left_right
      const float xmin = __fsub_rn(1.0f, q.w), xmax = __fsub_rn(1.0f, q.y);
      q.y = xmin; q.w = xmax;
    }
    q.x = fminf(fmaxf(__fsub_rn(__fmul_rn(sh, q.x), oy), 0.f), hy);
    q.y = fminf(fmaxf(__fsub_rn(__fmul_rn(sw, q.y), ox), 0.f), hx);
    q.z = fminf(fmaxf(__fsub_rn(__fmul_rn(sh, q.z), oy), 0.f), hy);
    q.w = fminf(fmaxf(__fsub_rn(__fmul_rn(sw, q.w), ox), 0.f), hx);
    tb[i] = q;
    keep[i] = __fmul_rn(__fsub_rn(q.z, q.x), __fsub_rn(q.w, q.y)) != 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (keep[i]) keep[m++] = i;                     // m <= i: in-place list of the kept rows
    counts_out[b] = m;
    kept = m;
  }
  __syncthreads();
  const int m = kept;
  for (int i = tid; i < max_boxes; i += blockDim.x) {
    float4 q = make_float4(-1.f, -1.f, -1.f, -1.f);
    float c = -1.f;
    if (i < m) {
      q = tb[keep[i]];
      c = classes_in[(size_t)b * max_boxes + keep[i]];
    }
    *reinterpret_cast<float4*>(boxes_out + ((size_t)b * max_boxes + i) * 4) = q;
    classes_out[(size_t)b * max_boxes + i] = c;
  }
}

}  // namespace

extern "C" int edet_preprocess_infer(const void* raw_images, int raw_is_float, int batch, int height, int width,
                                     int out_height, int out_width, const float* mean_rgb, const float* stddev_rgb,
                                     void* out, float* image_scale_to_original, int dtype, void* stream) {
  EDET_CHECK(raw_images && out && mean_rgb && stddev_rgb && image_scale_to_original, "edet_preprocess_infer: null");
  EDET_CHECK(batch >= 1 && height >= 1 && width >= 1 && out_height >= 1 && out_width >= 1,
             "edet_preprocess_infer: bad sizes");
  EDET_CHECK(dtype == EDET_F32 || dtype == EDET_BF16, "edet_preprocess_infer: dtype %d", dtype);
  PrepArgs a;
  a.raw = raw_images; a.raw_is_float = raw_is_float;
  a.batch = batch; a.h = height; a.w = width; a.out_h = out_height; a.out_w = out_width;
  // set_scale_factors_to_output_size (float32 arithmetic, int casts truncate)
  const float sy = (float)out_height / (float)height, sx = (float)out_width / (float)width;
  const float scale = fminf(sx, sy);
  a.scaled_h = (int)((float)height * scale);
  a.scaled_w = (int)((float)width * scale);
  EDET_CHECK(a.scaled_h >= 1 && a.scaled_w >= 1, "edet_preprocess_infer: the scaled image is empty");
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean_rgb[c]; a.stddev[c] = stddev_rgb[c]; }
  a.out = out;
  *image_scale_to_original = 1.0f / scale;       // HOST output: the same scale for every image of the batch
  const dim3 grid(cdiv(out_width, 64), cdiv(out_height, 4), batch);
  if (dtype == EDET_BF16) edet_launch(k_preprocess_infer<bf16_t>, grid, dim3(256), 0, to_stream(stream), a);
  else edet_launch(k_preprocess_infer<float>, grid, dim3(256), 0, to_stream(stream), a);
  EDET_LAUNCH_CHECK("edet_preprocess_infer");
  return 0;
}

extern "C" int edet_preprocess_train(const void* raw_images, int raw_is_float, int batch, int height, int width,
                                     int out_height, int out_width, const float* mean_rgb, const float* stddev_rgb,
                                     const edet_prep_image_t* per_image_dev, void* out, const float* boxes_in,
                                     const float* classes_in, const int* counts_in, int max_boxes, float* boxes_out,
                                     float* classes_out, int* counts_out, int dtype, void* stream) {
  EDET_CHECK(raw_images && out && mean_rgb && stddev_rgb && per_image_dev, "edet_preprocess_train: null");
  EDET_CHECK(batch >= 1 && height >= 1 && width >= 1 && out_height >= 1 && out_width >= 1,
             "edet_preprocess_train: bad sizes");
  EDET_CHECK(dtype == EDET_F32 || dtype == EDET_BF16, "edet_preprocess_train: dtype %d", dtype);
  PrepArgs a;
  a.raw = raw_images; a.raw_is_float = raw_is_float;
  a.batch = batch; a.h = height; a.w = width; a.out_h = out_height; a.out_w = out_width;
  a.scaled_h = a.scaled_w = 0;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean_rgb[c]; a.stddev[c] = stddev_rgb[c]; }
  a.out = out;
  const dim3 grid(cdiv(out_width, 64), cdiv(out_height, 4), batch);
  if (dtype == EDET_BF16) edet_launch(k_preprocess_train<bf16_t>, grid, dim3(256), 0, to_stream(stream), a, per_image_dev);
  else edet_launch(k_preprocess_train<float>, grid, dim3(256), 0, to_stream(stream), a, per_image_dev);
  if (max_boxes > 0) {
    EDET_CHECK(boxes_in && classes_in && counts_in && boxes_out && classes_out && counts_out,
               "edet_preprocess_train: null box arrays");
    EDET_CHECK(max_boxes <= PREP_MAX_BOXES, "edet_preprocess_train: max_boxes = %d (<= %d)", max_boxes, PREP_MAX_BOXES);
    edet_launch(k_preprocess_boxes, dim3(batch), dim3(256), 0, to_stream(stream), per_image_dev, boxes_in, classes_in, counts_in, max_boxes, out_height, out_width, boxes_out, classes_out, counts_out);
  }
  EDET_LAUNCH_CHECK("edet_preprocess_train");
  return 0;
}
