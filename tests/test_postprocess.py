"""Detection post-processing: oracle vs the executed reference code (CPU) and the HIP path vs
the oracle / the fixtures (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from automl_amd import anchors
from oracle import postprocess_oracle as porc

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'reference_postprocess.npz')
CASES = {'a': (128, 3, 7, 90), 'b': (96, 3, 5, 7)}     # tests/golden/make_golden_postprocess.py CASES
NMS_CONFIGS = {
    'gaussian': dict(method='gaussian', iou_thresh=None, score_thresh=0., sigma=None, pyfunc=False, max_nms_inputs=0,
                     max_output_size=100),
    'hard': dict(method='hard', iou_thresh=None, score_thresh=None, sigma=None, pyfunc=False, max_nms_inputs=0,
                 max_output_size=20),
    'gaussian_topk': dict(method='gaussian', iou_thresh=None, score_thresh=0.01, sigma=0.3, pyfunc=False,
                          max_nms_inputs=300, max_output_size=50),
    'linear': dict(method='linear', iou_thresh=0.4, score_thresh=0.02, sigma=None, pyfunc=True, max_nms_inputs=0,
                   max_output_size=30),
}


def load_case(g, cname, nname):
  size, lo, hi, ncls = CASES[cname]
  params = dict(min_level=lo, max_level=hi, aspect_ratios=[1.0, 2.0, 0.5], num_scales=3, anchor_scale=4.0,
                image_size=size, num_classes=ncls, data_format='channels_last', nms_configs=dict(NMS_CONFIGS[nname]))
  cls = [g['%s/cls_%d' % (cname, i)] for i in range(hi - lo + 1)]
  box = [g['%s/box_%d' % (cname, i)] for i in range(hi - lo + 1)]
  anchor_boxes = anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size).boxes
  return params, cls, box, np.asarray(anchor_boxes, np.float32), g['%s/scales' % cname], g['%s/ids' % cname]


def close(a, b, tol=2e-6):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, (a.shape, b.shape)
  return float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))


@pytest.mark.parametrize('cname', sorted(CASES))
@pytest.mark.parametrize('nname', sorted(NMS_CONFIGS))
def test_oracle_equals_the_executed_reference_postprocess(cname, nname):
  """tests/golden/reference_postprocess.npz: pre_nms, postprocess_global, postprocess_per_class and generate_detections
  of tf2/postprocess.py executed on the stand-in, the numpy-NMS variant through the REAL nms_np.per_class_nms."""
  g = np.load(GOLDEN)
  params, cls, box, anc, scales, ids = load_case(g, cname, nname)
  key = '%s/%s/' % (cname, nname)
  b, s, c = porc.pre_nms(params, cls, box, anc)
  assert np.array_equal(c, g[key + 'pre_classes'])
  assert close(b, g[key + 'pre_boxes']) and close(s, g[key + 'pre_scores'])
  if params['nms_configs']['method'] in ('gaussian', 'hard'):
    for fn, tag in ((porc.postprocess_global, 'global'), (porc.postprocess_per_class, 'per_class')):
      r = fn(params, cls, box, anc, scales)
      for nm, v in zip(('boxes', 'scores', 'classes', 'valid'), r):
        assert close(v, g[key + tag + '_' + nm]), (tag, nm)
    assert close(porc.generate_detections(params, cls, box, anc, scales, ids, False), g[key + 'det_tf'])
  params['nms_configs']['pyfunc'] = True
  for flip in (False, True):
    got = porc.generate_detections(params, cls, box, anc, scales, ids, flip)
    assert close(got, g[key + 'det_np_%d' % flip], 1e-5), flip


@pytest.mark.parametrize('method,cfg', [
    ('hard', dict(method='hard', iou_thresh=0.45, score_thresh=None, sigma=None)),
    ('gaussian', dict(method='gaussian', iou_thresh=None, score_thresh=None, sigma=None)),
    ('linear', dict(method='linear', iou_thresh=None, score_thresh=0.05, sigma=None))])
def test_numpy_nms_restatement_equals_nms_np(method, cfg):
  """oracle np_nms vs the outputs of the reference's own nms_np.nms on 12 clusters of 10 boxes: bit for bit."""
  g = np.load(GOLDEN)
  got = porc.np_nms(g['np/dets'].copy(), cfg)
  assert got.shape == g['np/' + method].shape and np.array_equal(got, g['np/' + method])


def test_nms_v5_properties():
  """The restated NonMaxSuppressionV5: hard mode never keeps two boxes above the IoU threshold, scores come out in
  non-increasing order, soft mode with a tiny sigma degenerates to picking the isolated boxes first, padding."""
  rng = np.random.default_rng(3)
  ctr = rng.uniform(0, 100, (40, 2)).repeat(5, 0) + rng.normal(0, 2, (200, 2))
  wh = rng.uniform(8, 20, (200, 2))
  boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
  scores = rng.uniform(0.05, 1, 200).astype(np.float32)
  idx, sc, valid = porc.nms_v5(boxes, scores, 50, 0.5, 0.0, 0.0, True)
  assert idx.shape == (50,) and np.all(np.diff(sc[:valid]) <= 0)
  for i in range(valid):
    for j in range(i):
      assert porc._iou_tf(boxes, idx[i], idx[j]) <= 0.5
  assert np.all(idx[valid:] == 0) and np.all(sc[valid:] == 0)
  idx2, sc2, valid2 = porc.nms_v5(boxes, scores, 50, 0.5, 0.001, 0.25, False)
  assert valid2 == 50 and idx2.shape == (50,) and np.all(np.diff(sc2) <= 0) and idx2[0] == np.argmax(scores)
  assert len(set(idx2.tolist())) == 50


# ------------------------------------------------------------------------------------------------ HIP path
def _cuda(arrs, dtype=torch.float32):
  return [torch.from_numpy(a).cuda().to(dtype) for a in arrs]


def _np(t):
  return t.detach().float().cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('cname', sorted(CASES))
@pytest.mark.parametrize('nname', sorted(NMS_CONFIGS))
def test_device_equals_the_executed_reference_postprocess(cname, nname):
  """The HIP path against the outputs of the reference's own postprocess.py / nms_np.py (fixtures), fp32 inputs:
  classes, indices and valid lengths exactly; scores to 1e-5; boxes to 1e-4 of the image size (expf and the order
  in which the decay factors of soft-NMS are multiplied differ in the last bit)."""
  from automl_amd import postprocess as pp
  g = np.load(GOLDEN)
  params, cls, box, anc, scales, ids = load_case(g, cname, nname)
  key = '%s/%s/' % (cname, nname)
  size = float(params['image_size'])
  b, s, c = pp.pre_nms(params, _cuda(cls), _cuda(box))
  assert np.array_equal(_np(c).astype(np.int32), g[key + 'pre_classes'])
  assert np.abs(_np(b) - g[key + 'pre_boxes']).max() <= 1e-4 * size
  assert np.abs(_np(s) - g[key + 'pre_scores']).max() <= 1e-6

  def check(got, want_boxes, want_scores, want_classes, want_valid, tag):
    nb, ns, nc, nv = got
    assert np.array_equal(_np(nv).astype(np.int32), np.asarray(want_valid, np.int32)), tag
    assert np.array_equal(_np(nc), want_classes), tag
    assert np.abs(_np(ns) - want_scores).max() <= 1e-5, tag
    assert np.abs(_np(nb) - want_boxes).max() <= 1e-4 * size * float(scales.max()), tag

  if params['nms_configs']['method'] in ('gaussian', 'hard'):
    for fn, tag in ((pp.postprocess_global, 'global'), (pp.postprocess_per_class, 'per_class')):
      got = fn(params, _cuda(cls), _cuda(box), torch.from_numpy(scales))
      check(got, *[g[key + tag + '_' + nm] for nm in ('boxes', 'scores', 'classes', 'valid')], tag)
    det = pp.generate_detections(params, _cuda(cls), _cuda(box), scales, ids, False)
    assert np.abs(_np(det) - g[key + 'det_tf']).max() <= 1e-4 * size * float(scales.max())
  params['nms_configs']['pyfunc'] = True
  for flip in (False, True):
    det = _np(pp.generate_detections(params, _cuda(cls), _cuda(box), scales, ids, flip))
    want = g[key + 'det_np_%d' % flip]
    assert np.array_equal(det[..., 6], want[..., 6]) and np.array_equal(det[..., 0], want[..., 0]), flip
    assert np.abs(det[..., 5] - want[..., 5]).max() <= 1e-5
    assert np.abs(det[..., 1:5] - want[..., 1:5]).max() <= 1e-4 * size * float(scales.max())


def _random_levels(rng, batch, size, lo, hi, ncls, hot):
  sizes, s = [], size
  for level in range(1, hi + 1):
    s = (s - 1) // 2 + 1
    if level >= lo:
      sizes.append(s)
  cls, box = [], []
  for s in sizes:
    c = rng.normal(-3.0, 1.5, (batch, s, s, 9 * ncls)).astype(np.float32)
    m = rng.random(c.shape) < hot
    c[m] += rng.uniform(3.0, 7.0, int(m.sum())).astype(np.float32)
    cls.append(c)
    box.append(rng.normal(0.0, 0.25, (batch, s, s, 36)).astype(np.float32))
  return cls, box


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('nname,topk', [('gaussian', 0), ('hard', 0), ('gaussian', 5000), ('linear', 0)])
def test_device_matches_oracle_d0_256(dtype, nname, topk):
  """d0 levels 3-7 at 256 px (12,276 anchors x 90 classes, batch 2), fp32 and bf16 network outputs, with the eval
  configuration max_nms_inputs = 5000 (tf2/eval.py:52) among the cases.  bf16 logits tie often: the candidate
  order (first maximum class, lower flat index among equal top-k keys) must still be the reference's."""
  from automl_amd import postprocess as pp
  size, lo, hi, ncls, batch = 256, 3, 7, 90, 2
  rng = np.random.default_rng(77 + topk)
  cls, box = _random_levels(rng, batch, size, lo, hi, ncls, 0.0015)
  if dtype == torch.bfloat16:
    cls = [torch.from_numpy(c).to(torch.bfloat16).float().numpy() for c in cls]
    box = [torch.from_numpy(x).to(torch.bfloat16).float().numpy() for x in box]
  cfg = dict(NMS_CONFIGS[nname], max_nms_inputs=topk)
  params = dict(min_level=lo, max_level=hi, aspect_ratios=[1.0, 2.0, 0.5], num_scales=3, anchor_scale=4.0,
                image_size=size, num_classes=ncls, data_format='channels_last', nms_configs=cfg)
  anc = np.asarray(anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size).boxes, np.float32)
  scales = np.asarray([1.0, 1.7], np.float32)
  ids = np.asarray([3, 4])
  wb, ws_, wc = porc.pre_nms(params, cls, box, anc)
  b, s, c = pp.pre_nms(params, _cuda(cls, dtype), _cuda(box, dtype))
  assert np.array_equal(_np(c).astype(np.int32), wc)
  assert np.abs(_np(s) - ws_).max() <= 1e-6 and np.abs(_np(b) - wb).max() <= 1e-4 * size
  if cfg['method'] in ('gaussian', 'hard'):
    for fn, ofn in ((pp.postprocess_global, porc.postprocess_global), (pp.postprocess_per_class, porc.postprocess_per_class)):
      got = fn(params, _cuda(cls, dtype), _cuda(box, dtype), scales)
      want = ofn(params, cls, box, anc, scales)
      assert np.array_equal(_np(got[3]).astype(np.int32), want[3])
      assert np.array_equal(_np(got[2]), want[2])
      assert np.abs(_np(got[1]) - want[1]).max() <= 1e-5
      assert np.abs(_np(got[0]) - want[0]).max() <= 1e-4 * size * 1.7
  params['nms_configs']['pyfunc'] = True
  det = _np(pp.generate_detections(params, _cuda(cls, dtype), _cuda(box, dtype), scales, ids, False))
  want = porc.generate_detections(params, cls, box, anc, scales, ids, False)
  # rows of equal score (bf16 logits) come out of np.argsort(-scores) in no defined order: canonical row order
  canon = lambda d: np.stack([r[np.lexsort((r[:, 1], r[:, 6], -r[:, 5]))] for r in d])      # noqa: E731
  det, want = canon(det), canon(want)
  assert np.array_equal(det[..., 6], want[..., 6])
  assert np.abs(det[..., 5] - want[..., 5]).max() <= 1e-5
  assert np.abs(det[..., 1:5] - want[..., 1:5]).max() <= 1e-4 * size * 1.7


@pytest.mark.gpu
def test_device_properties_at_d0_640_batch_8():
  """BASELINE size (76,725 anchors x 90 classes per image, batch 8, bf16): size-independent properties -- the
  top-k branch returns exactly the k largest logits in order (checked against torch.topk), global hard NMS returns
  non-increasing scores, boxes inside the clip window, no pair above the IoU threshold, classes in 1..90."""
  from automl_amd import postprocess as pp
  size, lo, hi, ncls, batch = 640, 3, 7, 90, 8
  rng = np.random.default_rng(5)
  cls, box = _random_levels(rng, batch, size, lo, hi, ncls, 0.0005)
  tc, tb = _cuda(cls, torch.bfloat16), _cuda(box, torch.bfloat16)
  cfg = dict(NMS_CONFIGS['hard'], max_nms_inputs=5000, max_output_size=100)
  params = dict(min_level=lo, max_level=hi, aspect_ratios=[1.0, 2.0, 0.5], num_scales=3, anchor_scale=4.0,
                image_size=size, num_classes=ncls, data_format='channels_last', nms_configs=cfg)
  boxes, scores, classes = pp.pre_nms(params, tc, tb)
  flat = torch.cat([c.reshape(batch, -1) for c in tc], 1).float()
  top = torch.topk(flat, 5000, dim=1).values
  assert torch.equal(torch.sigmoid(top), scores) or float((torch.sigmoid(top) - scores).abs().max()) <= 1e-6
  assert bool((scores[:, 1:] <= scores[:, :-1]).all())
  nb, ns, nc, nv = pp.postprocess_global(params, tc, tb)
  raw = pp._run_nms(pp._tf_nms_cfg(cfg), boxes, scores, classes, 1, 0)[0]      # the same selection, unclipped boxes
  nb, ns, nc, nv, raw = _np(nb), _np(ns), _np(nc), _np(nv).astype(int), _np(raw)
  for i in range(batch):
    v = nv[i]
    assert 1 <= v <= 100 and np.all(np.diff(ns[i, :v]) <= 0)
    assert nb[i].min() >= 0 and nb[i].max() <= size and nc[i, :v].min() >= 1 and nc[i, :v].max() <= ncls
    assert np.array_equal(np.clip(raw[i], 0, size), nb[i])
    for p in range(v):
      for q in range(p):
        assert porc._iou_tf(raw[i], p, q) <= 0.5 + 1e-6


def test_model_wrapper_dispatch(monkeypatch):
  """EfficientDetModel (efficientdet_keras.py:917-1000): argument handling without a GPU -- the network call and the
  post-processing functions are replaced by recorders."""
  from automl_amd import efficientdet_net, hparams_config, postprocess as pp
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  model = efficientdet_net.EfficientDetModel(config=config)
  calls = []
  monkeypatch.setattr(efficientdet_net.EfficientDetNet, '__call__', lambda self, x, training=False: (['c'], ['b']))
  monkeypatch.setattr(pp, 'postprocess_global', lambda p, c, b, s=None: calls.append(('global', p['name'], c, b, s)) or 'G')
  monkeypatch.setattr(pp, 'postprocess_per_class', lambda p, c, b, s=None: calls.append(('per_class', c, b, s)) or 'P')
  from automl_amd import preprocess
  monkeypatch.setattr(preprocess, 'preprocess_infer', lambda raw, size, mean, std, dtype=None: ('prep', 'scales'))
  assert model('raw') == 'G' and calls[-1][-1] == 'scales'      # the reference default pre_mode='infer'
  with pytest.raises(ValueError, match='preprocessing must be infer or empty'):
    model(None, pre_mode='train')
  assert model(None, pre_mode=None) == 'G' and calls[-1] == ('global', 'efficientdet-d0', ['c'], ['b'], None)
  assert model(None, pre_mode=None, post_mode='per_class') == 'P'
  assert model(None, pre_mode=None, post_mode=None) == (['c'], ['b'])
  for mode in ('combined', 'tflite'):
    with pytest.raises(ValueError, match='not built'):
      model(None, pre_mode=None, post_mode=mode)
  with pytest.raises(ValueError, match='Unsupported postprocess mode'):
    model(None, pre_mode=None, post_mode='bogus')


@pytest.mark.gpu
def test_model_wrapper_equals_network_plus_postprocess():
  """EfficientDetModel(post_mode=...) == postprocess_*() of the level outputs of that very forward pass (two forward
  passes are not bit-identical: the SE pooling sums are fp32 atomics), fp32 storage."""
  from automl_amd import efficientdet_net, hparams_config, postprocess as pp
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=128')
  rng = np.random.default_rng(9)
  images = torch.from_numpy(rng.standard_normal((2, 128, 128, 3)).astype(np.float32))
  model = efficientdet_net.EfficientDetModel(config=config, dtype='f32', seed=3)
  for mode, fn in (('global', pp.postprocess_global), ('per_class', pp.postprocess_per_class)):
    got = model(images, training=False, pre_mode=None, post_mode=mode)
    cls, box = model.engine.outputs()                       # the logits that call has just post-processed
    want = fn(config.as_dict(), [c.clone() for c in cls], [b.clone() for b in box])
    assert len(got) == 4 and tuple(got[0].shape) == (2, 100, 4) and tuple(got[3].shape) == (2,)
    for g, w in zip(got, want):
      assert torch.equal(g, w), mode
  raw = model(images, pre_mode=None, post_mode=None)
  assert len(raw[0]) == 5 and tuple(raw[0][0].shape) == (2, 16, 16, 810) and tuple(raw[1][4].shape) == (2, 1, 1, 36)
