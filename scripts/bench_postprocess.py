"""Timing of the detection post-processing kernels at the BASELINE size (EfficientDet-D0 640x640: 76,725 anchors x 90
classes per image), bf16 network outputs resident in HBM.  Prints one JSON line per entry point:
algorithmic bytes = logits + box codes read once + outputs written (edet_pre_nms), HIP-event time per call.

usage: python scripts/bench_postprocess.py [--batch 128] [--reps 10]
"""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import postprocess as pp   # noqa


def timed(fn, reps):
  fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--reps', type=int, default=10)
  args = ap.parse_args()
  size, lo, hi, ncls, b = 640, 3, 7, 90, args.batch
  gen = torch.Generator(device='cuda').manual_seed(1)
  cls, box, s = [], [], size
  for level in range(1, hi + 1):
    s = (s - 1) // 2 + 1
    if level < lo:
      continue
    c = torch.randn((b, s, s, 9 * ncls), device='cuda', generator=gen) * 1.5 - 3.0
    hot = torch.rand(c.shape, device='cuda', generator=gen) < 0.0005      # a few hundred confident anchors per image
    c = torch.where(hot, c + 6.0, c)
    cls.append(c.to(torch.bfloat16))
    box.append((torch.randn((b, s, s, 36), device='cuda', generator=gen) * 0.25).to(torch.bfloat16))
  n = sum(c.shape[1] * c.shape[2] * 9 for c in cls)

  def params(method, topk, m=100):
    return dict(min_level=lo, max_level=hi, aspect_ratios=[1.0, 2.0, 0.5], num_scales=3, anchor_scale=4.0,
                image_size=size, num_classes=ncls, data_format='channels_last',
                nms_configs=dict(method=method, iou_thresh=None, score_thresh=None, sigma=None, pyfunc=False,
                                 max_nms_inputs=topk, max_output_size=m))
  rows = []
  p0 = params('gaussian', 0)
  ms = timed(lambda: pp.pre_nms(p0, cls, box), args.reps)
  nbytes = b * n * (ncls * 2 + 4 * 2 + 16 + 4 + 4) + n * 16
  rows.append(dict(entry='edet_pre_nms', ms=ms, algorithmic_bytes=nbytes, achieved_GBps=nbytes / ms / 1e6,
                   hbm_frac=nbytes / ms / 1e6 / 8000.0))
  p1 = params('gaussian', 5000)
  ms = timed(lambda: pp.pre_nms(p1, cls, box), args.reps)
  rows.append(dict(entry='edet_pre_nms_topk(k=5000)', ms=ms, algorithmic_bytes=b * n * ncls * 2,
                   achieved_GBps=b * n * ncls * 2 / ms / 1e6, note='5 passes over the logits (3 radix digits, 2 collects)'))
  for name, fn, p in (('postprocess_global gaussian (all %d candidates)' % n, pp.postprocess_global, p0),
                      ('postprocess_global gaussian (top 5000)', pp.postprocess_global, p1),
                      ('postprocess_per_class gaussian (top 5000)', pp.postprocess_per_class, p1),
                      ('postprocess_global hard (top 5000)', pp.postprocess_global, params('hard', 5000))):
    ms = timed(lambda: fn(p, cls, box), max(2, args.reps // 3))
    rows.append(dict(entry=name, ms=ms, images_per_sec=b / ms * 1e3))
  pn = params('gaussian', 5000)
  pn['nms_configs']['pyfunc'] = True
  ids, scales = np.arange(b), np.ones(b, np.float32)
  ms = timed(lambda: pp.generate_detections(pn, cls, box, scales, ids), max(2, args.reps // 3))
  rows.append(dict(entry='generate_detections numpy-NMS semantics (top 5000)', ms=ms, images_per_sec=b / ms * 1e3))
  for r in rows:
    r.update(batch=b, anchors=n, dtype='bf16')
    print(json.dumps(r))


if __name__ == '__main__':
  main()
