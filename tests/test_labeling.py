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
