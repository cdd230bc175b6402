"""CPU oracle of the image / box preprocessing of the input pipeline -- TEST INFRASTRUCTURE.

Only tests/ (and the golden generators under tests/golden/) may import this module.  Plain numpy (float32)
restatement of (paths relative to the reference root):

  efficientdet/dataloader.py   InputProcessor.normalize_image :58-64, set_training_random_scale_factors :66-111,
                               set_scale_factors_to_output_size :113-124, resize_and_crop_image :126-139,
                               DetectionInputProcessor.random_horizontal_flip :150-153, clip_boxes :155-163,
                               resize_and_crop_boxes :165-189, image_scale_to_original :197-200
  efficientdet/object_detection/preprocessor.py   _flip_boxes_left_right :47-63, random_horizontal_flip :113-199,
                               box_list_scale :369-392
  efficientdet/tf2/efficientdet_keras.py   EfficientDetModel._preprocessing :920-951 (mode 'infer')

The random draws (flip decision, scale factor, crop offsets) are INPUTS: TF's RNG stream cannot be reproduced.
tf.image.resize (bilinear, TF2 half-pixel centres, no antialiasing) and tf.image.pad_to_bounding_box live in the
un-vendored tensorflow>=2.10,<2.16: `resize_bilinear` restates resize_bilinear_op.cc / image_resizer_state.h
(HalfPixelScaler: in = (out + 0.5) * in_size / out_size - 0.5, lower = max(floor(in), 0), upper = min(ceil(in),
in_size - 1), lerp = in - floor(in)).  PARITY STATUS: the glue is pinned by executing the reference classes on the
stand-in (tests/golden/make_golden_preprocess.py), where tf.image.resize is torch's bilinear interpolation
(align_corners=False, no antialias: the same published sampling rule, an independent implementation); parity of the
resize against the TensorFlow binary is UNPINNED.
"""
import numpy as np

F = np.float32


def normalize_image(image, mean_rgb, stddev_rgb):
  image = np.asarray(image).astype(np.float32)
  return (image - np.asarray(mean_rgb, np.float32).reshape(1, 1, 3)) / np.asarray(stddev_rgb, np.float32).reshape(1, 1, 3)


def scale_factors_to_output_size(height, width, output_size):
  """set_scale_factors_to_output_size -> (image_scale float32, scaled_height, scaled_width)."""
  h, w = F(height), F(width)
  scale = min(F(output_size[1]) / w, F(output_size[0]) / h)
  return F(scale), int(F(h * scale)), int(F(w * scale))


def training_random_scale_factors(height, width, output_size, target_size, random_scale_factor, u_y, u_x):
  """set_training_random_scale_factors with the three uniform draws as inputs
  -> (image_scale, scaled_height, scaled_width, offset_y, offset_x)."""
  target_size = target_size or output_size
  r = F(random_scale_factor)
  scaled_y, scaled_x = int(F(r * F(target_size[0]))), int(F(r * F(target_size[1])))
  h, w = F(height), F(width)
  scale = min(F(scaled_x) / w, F(scaled_y) / h)
  sh, sw = int(F(h * scale)), int(F(w * scale))
  oy = max(F(0), F(sh - output_size[0])) * F(u_y)
  ox = max(F(0), F(sw - output_size[1])) * F(u_x)
  return F(scale), sh, sw, int(F(oy)), int(F(ox))


def resize_bilinear(image, out_h, out_w):
  """tf.image.resize(method=BILINEAR) of an [H,W,C] float32 image (TF2: half-pixel centres, antialias=False)."""
  image = np.asarray(image, np.float32)
  in_h, in_w = image.shape[:2]

  def taps(out_n, in_n):
    scale = F(in_n) / F(out_n)
    src = (np.arange(out_n, dtype=np.float32) + F(0.5)) * scale - F(0.5)
    fl = np.floor(src)
    lo = np.maximum(fl, 0).astype(np.int64)
    hi = np.minimum(np.ceil(src), in_n - 1).astype(np.int64)
    return lo, hi, (src - fl).astype(np.float32)
  ylo, yhi, yl = taps(out_h, in_h)
  xlo, xhi, xl = taps(out_w, in_w)
  top = image[ylo][:, xlo] + (image[ylo][:, xhi] - image[ylo][:, xlo]) * xl[None, :, None]
  bot = image[yhi][:, xlo] + (image[yhi][:, xhi] - image[yhi][:, xlo]) * xl[None, :, None]
  return (top + (bot - top) * yl[:, None, None]).astype(np.float32)


def resize_and_crop_image(image, scaled_h, scaled_w, offset_y, offset_x, output_size):
  """resize -> crop window [offset, offset + output_size) -> zero pad at the bottom / right to output_size."""
  scaled = resize_bilinear(image, scaled_h, scaled_w)
  crop = scaled[offset_y:offset_y + output_size[0], offset_x:offset_x + output_size[1], :]
  out = np.zeros((output_size[0], output_size[1], image.shape[2]), np.float32)
  out[:crop.shape[0], :crop.shape[1]] = crop
  return out


def flip_left_right(image, boxes):
  """preprocessor.random_horizontal_flip with the decision taken (boxes normalised to [0, 1])."""
  b = np.asarray(boxes, np.float32)
  return np.asarray(image)[:, ::-1], np.stack([b[:, 0], F(1.0) - b[:, 3], b[:, 2], F(1.0) - b[:, 1]], 1)


def resize_and_crop_boxes(boxes, classes, scaled_h, scaled_w, offset_y, offset_x, output_size):
  """Normalised boxes -> pixels of the scaled image, minus the crop offset, clipped to [0, size - 1], zero-area
  boxes removed -> (boxes [K,4], classes [K,...])."""
  b = np.asarray(boxes, np.float32).reshape(-1, 4)
  b = b * np.asarray([scaled_h, scaled_w, scaled_h, scaled_w], np.float32)
  b = b - np.asarray([offset_y, offset_x, offset_y, offset_x], np.float32)
  hi = np.asarray([output_size[0] - 1, output_size[1] - 1, output_size[0] - 1, output_size[1] - 1], np.float32)
  b = np.clip(b, F(0), hi)
  keep = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])) != 0
  return b[keep], np.asarray(classes)[keep]


def preprocess_infer(raw_images, image_size, mean_rgb, stddev_rgb):
  """EfficientDetModel._preprocessing(mode='infer') -> (images [B,H,W,3], image_scales [B])."""
  images, scales = [], []
  for img in raw_images:
    x = normalize_image(img, mean_rgb, stddev_rgb)
    s, sh, sw = scale_factors_to_output_size(x.shape[0], x.shape[1], image_size)
    images.append(resize_and_crop_image(x, sh, sw, 0, 0, image_size))
    scales.append(F(1.0) / s)
  return np.stack(images), np.asarray(scales, np.float32)
