"""Image preprocessing on the GPU: ``EfficientDetModel._preprocessing(mode='infer')`` (efficientdet_keras.py:920-951) and
the training-time ``DetectionInputProcessor`` of the input pipeline (dataloader.py:144-200) for a batch."""
import ctypes

import numpy as np
import torch

from automl_amd import _lib
from automl_amd import utils


def preprocess_infer(raw_images, image_size, mean_rgb, stddev_rgb, dtype=torch.float32):
  """raw_images [B,H,W,3] uint8 or float32 (any device) -> (images [B,h,w,3] `dtype` on the GPU, image_scales [B]
  float32): normalised, resized with the aspect ratio kept into the top-left corner of image_size, zero padded."""
  if raw_images.dim() != 4 or raw_images.shape[-1] != 3:
    raise ValueError('raw images must be [batch, height, width, 3], got %s' % (tuple(raw_images.shape),))
  if raw_images.dtype not in (torch.uint8, torch.float32):
    raw_images = raw_images.to(torch.float32)
  if dtype not in (torch.float32, torch.bfloat16):
    raise ValueError('dtype must be float32 or bfloat16')
  raw = raw_images.cuda().contiguous()
  oh, ow = utils.parse_image_size(image_size)
  b, h, w, _ = raw.shape
  out = torch.empty((b, oh, ow, 3), dtype=dtype, device=raw.device)
  # per-channel lists, or one scalar for all channels (the efficientdet-lite configurations: 127.0 / 128.0)
  mean = (ctypes.c_float * 3)(*[float(v) for v in np.broadcast_to(np.asarray(mean_rgb, np.float32).reshape(-1), (3,))])
  std = (ctypes.c_float * 3)(*[float(v) for v in np.broadcast_to(np.asarray(stddev_rgb, np.float32).reshape(-1), (3,))])
  scale = ctypes.c_float(0.0)
  _lib.call('edet_preprocess_infer', raw.data_ptr(), 1 if raw.dtype == torch.float32 else 0, b, h, w, oh, ow, mean,
            std, out.data_ptr(), ctypes.byref(scale), _lib.EDET_BF16 if dtype == torch.bfloat16 else _lib.EDET_F32,
            torch.cuda.current_stream().cuda_stream)
  return out, torch.full((b,), scale.value, dtype=torch.float32, device=raw.device)


F = np.float32


MAX_BOXES = 1024      # PREP_MAX_BOXES of csrc/preprocess.hip


class DetectionInputProcessor(object):
  """``dataloader.DetectionInputProcessor`` (dataloader.py:144-200, base class :36-141) for a BATCH of equally sized raw
  images, same method names and call order as ``InputReader.process_example`` uses them (:321-336)::

      p = DetectionInputProcessor(raw_images, image_size, boxes, classes, counts)
      p.normalize_image(mean_rgb, stddev_rgb)
      p.random_horizontal_flip()                                   # training only
      p.set_training_random_scale_factors(jitter_min, jitter_max, target_size)   # or set_scale_factors_to_output_size()
      images = p.resize_and_crop_image()
      boxes, classes, counts = p.resize_and_crop_boxes()

  The setters only record per-image decisions (host, float32 arithmetic statement by statement as the reference's
  TensorFlow expressions: the int casts truncate); all pixel and box work is ONE call of edet_preprocess_train when the
  first result is asked for.  The random draws come from `rng` (numpy Generator) or are handed in (`draws=`), which is
  how the tests feed the values the reference's fixtures were made with -- TensorFlow's RNG stream cannot be
  reproduced.  boxes [B, max_boxes, 4] normalised (ymin, xmin, ymax, xmax), classes [B, max_boxes] (or [.., 1]), counts
  [B] valid rows per image (None: all rows)."""

  def __init__(self, images, output_size, boxes=None, classes=None, counts=None, rng=None, dtype=torch.float32):
    if images.dim() != 4 or images.shape[-1] != 3:
      raise ValueError('raw images must be [batch, height, width, 3], got %s' % (tuple(images.shape),))
    if dtype not in (torch.float32, torch.bfloat16):
      raise ValueError('dtype must be float32 or bfloat16')
    if images.dtype not in (torch.uint8, torch.float32):
      images = images.to(torch.float32)
    self._raw = images.cuda().contiguous()
    self._output_size = utils.parse_image_size(output_size)
    self._batch, self._height, self._width = (int(v) for v in self._raw.shape[:3])
    self._dtype = dtype
    self._rng = rng if rng is not None else np.random.default_rng()
    self._mean = self._std = None
    b = self._batch
    self._flip = np.zeros(b, np.int32)
    self._image_scale = np.ones(b, np.float32)
    self._scaled = np.zeros((b, 2), np.int32)
    self._offset = np.zeros((b, 2), np.int32)
    self._scales_set = False
    self._boxes = self._classes = self._counts = None
    if boxes is not None:
      self._boxes = torch.as_tensor(boxes, dtype=torch.float32).reshape(b, -1, 4).cuda().contiguous()
      m = int(self._boxes.shape[1])
      if m > MAX_BOXES:
        raise ValueError('at most %d boxes per image (edet_preprocess_train keeps them in LDS), got %d' % (MAX_BOXES, m))
      if classes is None:      # the reference's constructor allows it (dataloader.py:146): boxes without labels
        classes = torch.zeros(b, m, dtype=torch.float32)
      self._classes = torch.as_tensor(classes, dtype=torch.float32).reshape(b, m).cuda().contiguous()
      self._counts = (torch.full((b,), m, dtype=torch.int32) if counts is None
                      else torch.as_tensor(counts).to(torch.int32)).cuda().contiguous()
    self._result = None

  def normalize_image(self, mean_rgb, stddev_rgb):
    """Records (image - mean) / stddev, applied to the taps of the resize as the reference normalises first (:58-64)."""
    self._mean = np.broadcast_to(np.asarray(mean_rgb, np.float32).reshape(-1), (3,)).copy()
    self._std = np.broadcast_to(np.asarray(stddev_rgb, np.float32).reshape(-1), (3,)).copy()

  def _uniform(self, draws, count):
    if draws is None:
      return self._rng.random((self._batch, count)).astype(np.float32)
    d = np.asarray(draws, np.float32).reshape(self._batch, count)
    return d

  def random_horizontal_flip(self, draws=None):
    """preprocessor.random_horizontal_flip (object_detection/preprocessor.py:113-199): one uniform draw per image,
    flipped when it is > 0.5; image and boxes together."""
    self._flip = (self._uniform(draws, 1)[:, 0] > F(0.5)).astype(np.int32)
    self._result = None

  def set_training_random_scale_factors(self, scale_min, scale_max, target_size=None, draws=None):
    """dataloader.py:66-111; draws [B, 3] = the uniform [0, 1) values behind (scale factor, offset y, offset x)."""
    target = utils.parse_image_size(target_size) if target_size else self._output_size
    u = self._uniform(draws, 3)
    oh, ow = self._output_size
    for i in range(self._batch):
      factor = F(scale_min) + u[i, 0] * (F(scale_max) - F(scale_min))        # tf.random.uniform([], min, max)
      scaled_y, scaled_x = int(F(factor * F(target[0]))), int(F(factor * F(target[1])))
      h, w = F(self._height), F(self._width)
      scale = min(F(scaled_x) / w, F(scaled_y) / h)
      sh, sw = int(F(h * scale)), int(F(w * scale))
      oy = max(F(0), F(sh - oh)) * u[i, 1]
      ox = max(F(0), F(sw - ow)) * u[i, 2]
      self._image_scale[i], self._scaled[i], self._offset[i] = scale, (sh, sw), (int(F(oy)), int(F(ox)))
    self._scales_set = True
    self._result = None

  def set_scale_factors_to_output_size(self):
    """dataloader.py:113-124 (evaluation: the whole image into the top-left corner, no offset)."""
    h, w = F(self._height), F(self._width)
    scale = min(F(self._output_size[1]) / w, F(self._output_size[0]) / h)
    self._image_scale[:] = scale
    self._scaled[:] = (int(F(h * scale)), int(F(w * scale)))
    self._offset[:] = 0
    self._scales_set = True
    self._result = None

  def _run(self):
    if self._result is not None:
      return self._result
    if self._mean is None or not self._scales_set:
      raise RuntimeError('call normalize_image and one of the set_*scale_factors methods first')
    if int(self._scaled.min()) < 1:
      raise ValueError('the scaled image is empty')
    b = self._batch
    oh, ow = self._output_size
    per = np.concatenate([self._flip[:, None], self._scaled, self._offset], 1).astype(np.int32)
    per_dev = torch.from_numpy(np.ascontiguousarray(per)).cuda()
    out = torch.empty((b, oh, ow, 3), dtype=self._dtype, device=self._raw.device)
    mean, std = (ctypes.c_float * 3)(*[float(v) for v in self._mean]), (ctypes.c_float * 3)(*[float(v) for v in self._std])
    m = 0 if self._boxes is None else int(self._boxes.shape[1])
    bo = co = cnt = None
    if m:
      bo = torch.empty_like(self._boxes)
      co = torch.empty_like(self._classes)
      cnt = torch.empty_like(self._counts)
    p = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    _lib.call('edet_preprocess_train', self._raw.data_ptr(), 1 if self._raw.dtype == torch.float32 else 0, b,
              self._height, self._width, oh, ow, mean, std, per_dev.data_ptr(), out.data_ptr(), p(self._boxes),
              p(self._classes), p(self._counts), m, p(bo), p(co), p(cnt),
              _lib.EDET_BF16 if self._dtype == torch.bfloat16 else _lib.EDET_F32,
              torch.cuda.current_stream().cuda_stream)
    self._keep_alive = per_dev
    self._result = (out, bo, co, cnt)
    return self._result

  def resize_and_crop_image(self):
    """dataloader.py:126-139 -> [B, H, W, 3] on the GPU."""
    return self._run()[0]

  def resize_and_crop_boxes(self):
    """dataloader.py:165-189 -> (boxes [B, max_boxes, 4] in pixels of the output image, classes [B, max_boxes],
    counts [B]): the boxes of non-zero area in their original order, padded with -1."""
    if self._boxes is None:
      raise ValueError('no boxes were given')
    _, bo, co, cnt = self._run()
    return bo, co, cnt

  @property
  def image_scale(self):
    """Scale from the original image to the scaled image, [B] float32 (dataloader.py:191-194)."""
    return torch.from_numpy(self._image_scale.copy())

  @property
  def image_scale_to_original(self):
    """dataloader.py:196-199."""
    return torch.from_numpy((F(1.0) / self._image_scale).astype(np.float32))

  @property
  def offset_x(self):
    return torch.from_numpy(self._offset[:, 1].copy())

  @property
  def offset_y(self):
    return torch.from_numpy(self._offset[:, 0].copy())

  @property
  def scaled_size(self):
    """[B, 2] (scaled_height, scaled_width)."""
    return torch.from_numpy(self._scaled.copy())
