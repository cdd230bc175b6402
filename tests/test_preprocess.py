"""Image / box preprocessing: the numpy oracle against the reference's own InputProcessor /
DetectionInputProcessor executed on the stand-in (tests/golden/make_golden_preprocess.py), and the device kernels
(edet_preprocess_infer, edet_preprocess_train behind automl_amd.preprocess) against those fixtures and the oracle."""
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


def _train_oracle(raw, boxes, classes, osize, tsize, draws, jitter):
  """One image through the oracle's training path with the given draws (flip, scale, u_y, u_x)."""
  image = porc.normalize_image(raw, MEAN, STD)
  flip_u, scale_u, uy, ux = (np.float32(v) for v in draws)
  if flip_u > 0.5:
    image, boxes = porc.flip_left_right(image, boxes)
  factor = np.float32(jitter[0]) + scale_u * (np.float32(jitter[1]) - np.float32(jitter[0]))
  scale, sh, sw, oy, ox = porc.training_random_scale_factors(raw.shape[0], raw.shape[1], osize, tsize, factor, uy, ux)
  out = porc.resize_and_crop_image(image, sh, sw, oy, ox, osize)
  b, c = porc.resize_and_crop_boxes(boxes, classes, sh, sw, oy, ox, osize)
  return out, b, c, (scale, sh, sw, oy, ox)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['train_up_noflip', 'train_down_flip', 'train_crop_flip', 'infer_wide'])
def test_device_equals_the_executed_reference_detection_input_processor(name):
  """preprocess.DetectionInputProcessor (edet_preprocess_train) driven like InputReader.process_example
  (dataloader.py:321-336) with the fixtures' draws, against the outputs of the reference's own DetectionInputProcessor:
  scale / scaled size / crop offset exactly, image to 1e-4, kept boxes (scaled, shifted, clipped, zero-area filtered,
  in order) to 1e-5, classes exactly, padding rows -1."""
  import torch
  from automl_amd import preprocess
  g = np.load(GOLDEN)
  osize, tsize, training = CASES[name]
  raw, boxes, classes = g[name + '/raw'], g[name + '/boxes_in'], g[name + '/classes_in']
  pad = 3                                           # padded rows behind the valid ones must be ignored
  bpad = np.concatenate([boxes, np.full((pad, 4), 0.5, np.float32)])[None]
  cpad = np.concatenate([classes.reshape(-1), np.full((pad,), 77, np.float32)])[None]
  p = preprocess.DetectionInputProcessor(torch.from_numpy(raw)[None], osize, bpad, cpad, counts=[len(boxes)])
  p.normalize_image(MEAN, STD)
  if training:
    d = g[name + '/draws']
    p.random_horizontal_flip(draws=[d[0]])
    p.set_training_random_scale_factors(0.1, 2.0, tsize, draws=[d[1:4]])
  else:
    p.set_scale_factors_to_output_size()
  image = p.resize_and_crop_image()
  bo, co, cnt = p.resize_and_crop_boxes()
  torch.cuda.synchronize()
  assert p.scaled_size[0].tolist() + [int(p.offset_y[0]), int(p.offset_x[0])] == g[name + '/scaled'].tolist()
  assert np.float32(p.image_scale[0]) == g[name + '/image_scale']
  assert abs(float(p.image_scale_to_original[0]) - 1.0 / float(g[name + '/image_scale'])) <= 1e-6
  want = g[name + '/image']
  assert tuple(image.shape) == (1,) + want.shape and np.abs(image[0].cpu().numpy() - want).max() <= 1e-4
  k = int(cnt[0])
  assert k == len(g[name + '/boxes'])
  assert np.abs(bo[0, :k].cpu().numpy() - g[name + '/boxes']).max() <= 1e-5
  assert np.array_equal(co[0, :k].cpu().numpy(), g[name + '/classes'].reshape(-1))
  assert bool((bo[0, k:] == -1).all()) and bool((co[0, k:] == -1).all())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_device_training_preprocessing_matches_oracle_batch(dtype):
  """A batch of 8 frames, every image with its own flip / scale / crop draws (jitter 0.1 .. 2.0 as the reference's
  default hparams), uint8 and float inputs, against the oracle image by image; the processor's own random generator
  gives the same result as handing its draws in."""
  import torch
  from automl_amd import preprocess
  tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
  rng = np.random.default_rng(6)
  for (h, w, osize, tsize, as_float) in ((180, 240, (160, 160), None, False), (90, 70, (96, 128), (128, 128), True)):
    batch, nbox = 8, 12
    raw = rng.integers(0, 256, (batch, h, w, 3)).astype(np.uint8)
    ctr, hw = rng.uniform(0.1, 0.9, (batch, nbox, 2)), rng.uniform(0.02, 0.6, (batch, nbox, 2))
    boxes = np.clip(np.concatenate([ctr - hw / 2, ctr + hw / 2], 2), 0, 1).astype(np.float32)
    boxes[:, 0] = [0.3, 0.1, 0.3, 0.5]                                   # zero height: filtered everywhere
    classes = rng.integers(1, 91, (batch, nbox)).astype(np.float32)
    counts = rng.integers(3, nbox + 1, batch).astype(np.int32)
    draws = rng.random((batch, 4)).astype(np.float32)
    t = torch.from_numpy(raw.astype(np.float32)) if as_float else torch.from_numpy(raw)
    p = preprocess.DetectionInputProcessor(t, osize, boxes, classes, counts, dtype=tdt)
    p.normalize_image(MEAN, STD)
    p.random_horizontal_flip(draws=draws[:, :1])
    p.set_training_random_scale_factors(0.1, 2.0, tsize, draws=draws[:, 1:])
    image, (bo, co, cnt) = p.resize_and_crop_image(), p.resize_and_crop_boxes()
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == 'f32' else 2e-2
    flips = 0
    for i in range(batch):
      want, wb, wc, (scale, sh, sw, oy, ox) = _train_oracle(raw[i], boxes[i, :counts[i]], classes[i, :counts[i]], osize,
                                                           tsize, draws[i], (0.1, 2.0))
      flips += int(draws[i, 0] > 0.5)
      assert p.scaled_size[i].tolist() == [sh, sw] and [int(p.offset_y[i]), int(p.offset_x[i])] == [oy, ox]
      assert np.abs(image[i].float().cpu().numpy() - want).max() <= tol, (i, h, w)
      k = int(cnt[i])
      assert k == len(wb) and (k == 0 or np.abs(bo[i, :k].cpu().numpy() - wb).max() <= 1e-4)
      assert np.array_equal(co[i, :k].cpu().numpy(), wc)
      assert bool((bo[i, k:] == -1).all())
    assert 0 < flips < batch
    # the built-in generator: same seed -> the same draws -> the same tensors
    q = preprocess.DetectionInputProcessor(t, osize, boxes, classes, counts, rng=np.random.default_rng(99), dtype=tdt)
    q.normalize_image(MEAN, STD)
    q.random_horizontal_flip()
    q.set_training_random_scale_factors(0.1, 2.0, tsize)
    r = np.random.default_rng(99)
    d1, d3 = r.random((batch, 1)).astype(np.float32), r.random((batch, 3)).astype(np.float32)
    q2 = preprocess.DetectionInputProcessor(t, osize, boxes, classes, counts, dtype=tdt)
    q2.normalize_image(MEAN, STD)
    q2.random_horizontal_flip(draws=d1)
    q2.set_training_random_scale_factors(0.1, 2.0, tsize, draws=d3)
    assert torch.equal(q.resize_and_crop_image(), q2.resize_and_crop_image())
