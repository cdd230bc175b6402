"""Anchor labelling: the numpy oracle against the reference's own AnchorLabeler / object_detection
code executed on the stand-in (tests/golden/make_golden_labels.py).  The device kernels of this row are not built
yet; this file pins the oracle they will be checked against."""
import os

import numpy as np
import pytest

from automl_amd import anchors
from oracle import labeling_oracle as lorc

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'reference_labels.npz')
CASES = {'d0_256_8': (256, 3, 7), 'd0_256_0': (256, 3, 7), 'd0_384_40': (384, 3, 7), 'l8_320_5': (320, 3, 8),
         'dup_192': (192, 3, 7)}


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_equals_the_executed_reference_labeler(name):
  """Class targets (class - 1, -1 background) and the number of positives exactly; box targets to 1e-6 (same
  float32 formulas; log / divide may round differently between numpy and torch)."""
  g = np.load(GOLDEN)
  size, lo, hi = CASES[name]
  a = anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size)
  feat = [(a.feat_sizes[l]['height'], a.feat_sizes[l]['width']) for l in range(lo, hi + 1)]
  cls, box, npos = lorc.label_anchors(np.asarray(a.boxes, np.float32), feat, 9, g[name + '/gt_boxes'],
                                      g[name + '/gt_labels'])
  assert float(npos) == float(g[name + '/num_positives'])
  for i, level in enumerate(range(lo, hi + 1)):
    assert np.array_equal(cls[i], g['%s/cls_%d' % (name, level)]), level
    want = g['%s/box_%d' % (name, level)]
    assert box[i].shape == want.shape and np.abs(box[i] - want).max() <= 1e-6, level


def test_matcher_properties():
  """Every groundtruth box with positive area claims at least one anchor (force_match_for_each_row), anchors at or
  above the threshold are positive, encode() inverts anchors.decode_box_outputs."""
  rng = np.random.default_rng(0)
  a = anchors.Anchors(3, 7, 3, [1.0, 2.0, 0.5], 4.0, 256)
  an = np.asarray(a.boxes, np.float32)
  ctr = rng.uniform(30, 220, (12, 2))
  hw = rng.uniform(4, 120, (12, 2))
  gt = np.concatenate([ctr - hw / 2, ctr + hw / 2], 1).astype(np.float32)
  labels = rng.integers(1, 91, (12, 1))
  cls, reg, npos, match = lorc.label_anchors_flat(an, gt, labels)
  assert set(range(12)) <= set(match[match >= 0].tolist())
  sim = lorc.iou_matrix(gt, an)
  assert np.all(match[sim.max(0) >= 0.5] >= 0)
  assert npos == (match >= 0).sum() and np.all(cls[match < 0] == -1) and np.all(reg[match < 0] == 0)
  pos = match >= 0
  decoded = anchors.decode_box_outputs(reg[pos], an[pos])
  assert np.abs(np.asarray(decoded) - gt[match[pos]]).max() <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_device_equals_the_executed_reference_labeler(name):
  """edet_label_anchors against the fixtures of the executed reference code: class targets and positives exactly,
  box targets to 1e-6."""
  import torch
  from automl_amd import labeling
  g = np.load(GOLDEN)
  size, lo, hi = CASES[name]
  a = anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size)
  cls, box, npos = labeling.AnchorLabeler(a, 90).label_anchors(g[name + '/gt_boxes'], g[name + '/gt_labels'])
  torch.cuda.synchronize()
  assert float(npos) == float(g[name + '/num_positives'])
  for level in range(lo, hi + 1):
    assert np.array_equal(cls[level].cpu().numpy(), g['%s/cls_%d' % (name, level)]), level
    assert np.abs(box[level].cpu().numpy() - g['%s/box_%d' % (name, level)]).max() <= 1e-6, level


@pytest.mark.gpu
def test_device_batch_matches_oracle_d0_640():
  """BASELINE geometry (76,725 anchors), batch 16 with 0..100 boxes per image, against the oracle."""
  import torch
  from automl_amd import labeling
  rng = np.random.default_rng(21)
  size, lo, hi, b, mmax = 640, 3, 7, 16, 100
  a = anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size)
  an = np.asarray(a.boxes, np.float32)
  feat = [(a.feat_sizes[l]['height'], a.feat_sizes[l]['width']) for l in range(lo, hi + 1)]
  counts = rng.integers(0, mmax + 1, b)
  counts[0], counts[1] = 0, mmax
  gt = np.zeros((b, mmax, 4), np.float32) - 1
  labels = np.zeros((b, mmax), np.int32) - 1
  for i in range(b):
    ctr = rng.uniform(0.05, 0.95, (counts[i], 2)) * size
    hw = np.exp(rng.uniform(np.log(0.02), np.log(0.7), (counts[i], 2))) * size
    gt[i, :counts[i]] = np.clip(np.concatenate([ctr - hw / 2, ctr + hw / 2], 1), 0, size)
    labels[i, :counts[i]] = rng.integers(1, 91, counts[i])
  cls, box, npos = labeling.AnchorLabeler(a, 90).label_anchors_batch(gt, labels, counts)
  torch.cuda.synchronize()
  for i in range(b):
    wc, wb, wn = lorc.label_anchors(an, feat, 9, gt[i, :counts[i]], labels[i, :counts[i]])
    assert float(npos[i]) == float(wn), i
    for j, level in enumerate(range(lo, hi + 1)):
      assert np.array_equal(cls[level][i].cpu().numpy(), wc[j]), (i, level)
      assert np.abs(box[level][i].cpu().numpy() - wb[j]).max() <= 1e-6, (i, level)
