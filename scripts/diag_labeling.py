#!/usr/bin/env python3
"""Diagnostic: first differing anchors between edet_label_anchors and the oracle on the random d0-640 batch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import anchors, labeling  # noqa: E402
from oracle import labeling_oracle as lorc  # noqa: E402

rng = np.random.default_rng(21)
size, lo, hi, b, mmax = 640, 3, 7, 16, 100
a = anchors.Anchors(lo, hi, 3, [1.0, 2.0, 0.5], 4.0, size)
an = np.asarray(a.boxes, np.float32)
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
flat = np.concatenate([cls[l].cpu().numpy().reshape(b, -1) for l in range(lo, hi + 1)], 1)
shown = 0
for i in range(b):
  wc, wr, wn, match = lorc.label_anchors_flat(an, gt[i, :counts[i]], labels[i, :counts[i]])
  bad = np.nonzero(flat[i] != wc)[0]
  if len(bad) == 0:
    continue
  sim = lorc.iou_matrix(gt[i, :counts[i]], an)
  print('image %d: %d anchors differ, boxes %d' % (i, len(bad), counts[i]))
  for n in bad[:6]:
    col = sim[:, n]
    top = np.argsort(-col)[:3]
    print('  anchor %d dev cls %d oracle cls %d match %d | top IoUs %s rows %s labels-1 %s' % (
        n, flat[i, n], wc[n], match[n], [float.hex(float(col[t])) for t in top], top.tolist(),
        [int(labels[i, t]) - 1 for t in top]))
    m = match[n]
    if m >= 0:
      row = sim[m]
      best = np.flatnonzero(row == row.max())
      print('    oracle row %d: best anchors (ties) %s max %s; IoU(row, n) %s' % (m, best[:8].tolist(), float.hex(float(row.max())), float.hex(float(row[n]))))
  shown += 1
  if shown >= 3:
    break
