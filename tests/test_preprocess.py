"""Image / box preprocessing: the numpy oracle against the reference's own InputProcessor /
DetectionInputProcessor executed on the stand-in (tests/golden/make_golden_preprocess.py).  The device kernels of this
row are not built yet; this file pins the oracle they will be checked against."""
import os

import numpy as np
import pytest

from oracle import preprocess_oracle as porc

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'reference_preprocess.npz')
MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]
CASES = {   # name: (output size, target size, training)       -- tests/golden/make_golden_preprocess.py CASES
    'infer_wide': ((128, 128), None, False),
    'infer_tall': ((128, 160), None, False),
    'train_up_noflip': ((128, 128), None, True),
    'train_down_flip': ((128, 128), None, True),
    'train_crop_flip': ((96, 128), (128, 128), True),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_equals_the_executed_reference_input_processor(name):
  """Scale factors, scaled sizes and crop offsets exactly; images to 1e-4 (two independent float32 implementations
  of the half-pixel bilinear rule); boxes / classes after scale, crop, clip and the zero-area filter to 1e-5."""
  g = np.load(GOLDEN)
  osize, tsize, training = CASES[name]
  raw, boxes, classes = g[name + '/raw'], g[name + '/boxes_in'], g[name + '/classes_in']
  image = porc.normalize_image(raw, MEAN, STD)
  h, w = raw.shape[:2]
  if training:
    flip_u, scale_u, uy, ux = (np.float32(v) for v in g[name + '/draws'])
    if flip_u > 0.5:
      image, boxes = porc.flip_left_right(image, boxes)
    factor = np.float32(0.1) + scale_u * (np.float32(2.0) - np.float32(0.1))
    scale, sh, sw, oy, ox = porc.training_random_scale_factors(h, w, osize, tsize, factor, uy, ux)
  else:
    scale, sh, sw = porc.scale_factors_to_output_size(h, w, osize)
    oy = ox = 0
  assert [sh, sw, oy, ox] == g[name + '/scaled'].tolist()
  assert np.float32(scale) == g[name + '/image_scale']
  out = porc.resize_and_crop_image(image, sh, sw, oy, ox, osize)
  want = g[name + '/image']
  assert out.shape == want.shape and np.abs(out - want).max() <= 1e-4      # measured <= 2.9e-5 on values up to 2.6
  b, c = porc.resize_and_crop_boxes(boxes, classes, sh, sw, oy, ox, osize)
  assert b.shape == g[name + '/boxes'].shape and np.abs(b - g[name + '/boxes']).max() <= 1e-5
  assert np.array_equal(c, g[name + '/classes'])


def test_preprocess_infer_shapes_and_scales():
  """EfficientDetModel._preprocessing('infer'): aspect-preserving resize into the top-left corner, zero padding,
  image_scale = 1 / scale back to the original pixels."""
  rng = np.random.default_rng(0)
  raws = [rng.integers(0, 256, (60, 100, 3)).astype(np.uint8), rng.integers(0, 256, (100, 50, 3)).astype(np.uint8)]
  images, scales = porc.preprocess_infer(raws, (64, 64), MEAN, STD)
  assert images.shape == (2, 64, 64, 3) and np.allclose(scales, [100 / 64, 100 / 64])
  assert np.all(images[0, 39:] == 0) and np.all(images[1, :, 32:] == 0)       # 38 x 64 and 64 x 32 scaled images
  assert np.abs(images[0, :38]).max() > 0.5


def test_resize_bilinear_identity_and_constant():
  x = np.random.default_rng(1).standard_normal((7, 9, 3)).astype(np.float32)
  assert np.array_equal(porc.resize_bilinear(x, 7, 9), x)
  assert np.allclose(porc.resize_bilinear(np.full((5, 4, 1), 3.0, np.float32), 11, 13), 3.0)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['infer_wide', 'infer_tall'])
def test_device_equals_the_executed_reference_infer_preprocessing(name):
  """edet_preprocess_infer against the fixtures of the executed reference InputProcessor (fp32 output, 1e-4)."""
  import torch
  from automl_amd import preprocess
  g = np.load(GOLDEN)
  osize = CASES[name][0]
  out, scales = preprocess.preprocess_infer(torch.from_numpy(g[name + '/raw'])[None], osize, MEAN, STD)
  torch.cuda.synchronize()
  want = g[name + '/image']
  assert tuple(out.shape) == (1,) + want.shape and np.abs(out[0].cpu().numpy() - want).max() <= 1e-4
  assert abs(float(scales[0]) - 1.0 / float(g[name + '/image_scale'])) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_device_infer_preprocessing_matches_oracle_batch(dtype):
  """A batch of 1080p-like frames down to 640x640 and an upscaling case, uint8 and float inputs."""
  import torch
  from automl_amd import preprocess
  tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
  rng = np.random.default_rng(4)
  for (h, w, osize, as_float) in ((270, 480, (160, 160), False), (45, 80, (128, 96), True)):
    raw = rng.integers(0, 256, (3, h, w, 3)).astype(np.uint8)
    t = torch.from_numpy(raw.astype(np.float32)) if as_float else torch.from_numpy(raw)
    out, scales = preprocess.preprocess_infer(t, osize, MEAN, STD, dtype=tdt)
    want, wscales = porc.preprocess_infer(list(raw), osize, MEAN, STD)
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == 'f32' else 2e-2
    assert np.abs(out.float().cpu().numpy() - want).max() <= tol
    assert np.allclose(scales.cpu().numpy(), wscales, rtol=1e-6)


@pytest.mark.gpu
def test_model_call_with_infer_preprocessing_equals_preprocessed_call():
  """EfficientDetModel(raw images, pre_mode='infer', post_mode='global') (efficientdet_keras.py:978-1000) = the same
  model on the oracle-preprocessed batch with pre_mode=None, boxes scaled back by image_scale_to_original."""
  import torch
  from automl_amd import efficientdet_net, hparams_config
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=256')
  rng = np.random.default_rng(8)
  raw = rng.integers(0, 256, (2, 300, 420, 3)).astype(np.uint8)
  model = efficientdet_net.EfficientDetModel(config=config, dtype='f32')
  boxes, scores, classes, valid = model(torch.from_numpy(raw), training=False, pre_mode='infer', post_mode='global')
  images, scales = porc.preprocess_infer(list(raw), (256, 256), config.mean_rgb, config.stddev_rgb)
  b2, s2, c2, v2 = model(torch.from_numpy(images), training=False, pre_mode=None, post_mode='global')
  torch.cuda.synchronize()
  assert np.array_equal(valid.cpu().numpy(), v2.cpu().numpy())
  assert np.allclose(scores.cpu().numpy(), s2.cpu().numpy(), atol=2e-4)
  assert np.array_equal(classes.cpu().numpy(), c2.cpu().numpy())
  want = b2.cpu().numpy() * scales[:, None, None]
  assert np.abs(boxes.cpu().numpy() - want).max() <= 2e-3 * 420
  with pytest.raises(ValueError):
    model(torch.from_numpy(raw), pre_mode='train')
