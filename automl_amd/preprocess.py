"""Inference image preprocessing on the GPU: ``EfficientDetModel._preprocessing(mode='infer')``
(efficientdet_keras.py:920-951)."""
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
