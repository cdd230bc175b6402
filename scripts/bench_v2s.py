#!/usr/bin/env python3
"""BASELINE.json configs[1] (parity-test case, measured for DESIGN.md; NOT the bench.py line):
efficientnetv2-s backbone, 224x224, batch 256, bf16, forward (training=False) on one MI355X.
Prints one JSON line: images/s, ms per forward, per-entry-point kernel ms of one profiled forward."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from automl_amd import _lib, effnetv2_model  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--model', default='efficientnetv2-s')
  ap.add_argument('--batch', type=int, default=256)
  ap.add_argument('--size', type=int, default=224)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--dump_launches', default='')
  args = ap.parse_args()
  net = effnetv2_model.EffNetV2Model(args.model, include_top=False, dtype='bf16')
  rng = np.random.default_rng(2)
  images = torch.from_numpy(rng.standard_normal((args.batch, args.size, args.size, 3)).astype(np.float32))
  images = images.to('cuda:0', torch.bfloat16).contiguous()
  eng = net._ensure_engine(args.batch, args.size, args.size)
  for _ in range(args.warmup):
    eng.forward(images, training=False)
  torch.cuda.synchronize()
  _lib.profiler = _lib.Profiler(None)
  eng.forward(images, training=False)
  torch.cuda.synchronize()
  prof, by_shape = _lib.profiler.summary(), _lib.profiler.by_shape()
  _lib.profiler = None
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    eng.forward(images, training=False)
  g.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / args.steps
  if args.dump_launches:
    with open(args.dump_launches, 'w') as f:
      for (name, tag), (n, ms, b) in sorted(by_shape.items(), key=lambda kv: -kv[1][1]):
        f.write('%-18s %-26s %4d %9.3f %9.1f %8.1f\n' % (name, tag, n, ms, b / n / 1e6,
                                                       b / (ms * 1e-3) / 1e9 if ms > 0 else 0))
  alg = sum(v[2] for v in prof.values())
  print(json.dumps({
      'workload': '%s backbone %dx%d batch %d bf16 forward (inference BatchNorm), hipGraph replay'
                  % (args.model, args.size, args.size, args.batch),
      'images_per_sec': args.batch / dt, 'ms_per_forward': dt * 1e3,
      'kernel_ms_profiled': round(sum(v[1] for v in prof.values()), 3),
      'algorithmic_GB': round(alg / 1e9, 3), 'hbm_frac_8TBs': alg / dt / 8e12,
      'per_entry_ms': {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}}))


if __name__ == '__main__':
  main()
