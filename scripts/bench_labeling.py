#!/usr/bin/env python3
"""Anchor labelling on the device at the BASELINE size: EfficientDet-D0 640x640 = 76,725 anchors,
100 groundtruth boxes per image, batch 128.  HIP-event timing of AnchorLabeler.label_anchors_batch; prints one JSON
line (ms per call, images / s, IoU pairs / s)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import anchors, labeling  # noqa: E402


def main():
  size, b, m = 640, 128, 100
  a = anchors.Anchors(3, 7, 3, [1.0, 2.0, 0.5], 4.0, size)
  rng = np.random.default_rng(0)
  ctr = rng.uniform(0.05, 0.95, (b, m, 2)) * size
  hw = np.exp(rng.uniform(np.log(0.02), np.log(0.7), (b, m, 2))) * size
  gt = np.clip(np.concatenate([ctr - hw / 2, ctr + hw / 2], 2), 0, size).astype(np.float32)
  labels = rng.integers(1, 91, (b, m)).astype(np.int32)
  counts = np.full((b,), m, np.int32)
  lab = labeling.AnchorLabeler(a, 90)
  gtd, ld, cd = (torch.from_numpy(x).cuda() for x in (gt, labels, counts))
  for _ in range(3):
    out = lab.label_anchors_batch(gtd, ld, cd)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = 20
  s.record()
  for _ in range(reps):
    out = lab.label_anchors_batch(gtd, ld, cd)
  e.record()
  torch.cuda.synchronize()
  ms = s.elapsed_time(e) / reps
  n = int(np.asarray(a.boxes).shape[0])
  print(json.dumps({'workload': 'label_anchors_batch: %d anchors x %d boxes x %d images' % (n, m, b),
                    'ms_per_call': ms, 'images_per_sec': b / ms * 1e3,
                    'iou_pairs_per_sec': 2.0 * n * m * b / ms * 1e3,      # force-match pass + assignment pass
                    'mean_positives_per_image': float(out[2].mean())}))


if __name__ == '__main__':
  main()
