// Inference image preprocessing on device.
//
//   edet_preprocess_infer   efficientdet_keras.EfficientDetModel._preprocessing(mode='infer') :920-951 =
//                           dataloader.InputProcessor.normalize_image :58-64, set_scale_factors_to_output_size
//                           :113-124, resize_and_crop_image :126-139 (tf.image.resize bilinear with half-pixel
//                           centres, tf.image.pad_to_bounding_box) for a batch of equally sized raw images.
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
