#!/usr/bin/env python3
"""Diagnostic: where do a batch-N executor (N/2 copies of 2 images) and the 2-image executor diverge?  Prints, in
forward order, the error of every stored activation (then gradient) buffer; also two runs of the SAME executor
(atomics / ordering noise floor).  usage: diag_tiling.py [dtype] [big_batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_bench_shapes as tb  # noqa: E402


def compare(b2, bN, reps, label, every=12):
  rows = []
  for key, t2 in b2.items():
    tN = bN.get(key)
    if tN is None or t2.dim() != 4 or t2.shape[0] != 2 or tN.shape[0] != 2 * reps:
      continue
    a = tN.view((reps, 2) + tuple(t2.shape[1:])).float()
    b = t2.float().unsqueeze(0)
    is_grad = key.endswith('#grad') or key.endswith(':ds')
    if is_grad:
      a = a * reps
    ref = float(b.abs().max())
    if ref == 0 or not np.isfinite(ref):
      continue
    d = (a - b)
    rows.append((key, is_grad, float(d.abs().max()) / ref, float(d.pow(2).mean().sqrt()) / max(float(b.pow(2).mean().sqrt()), 1e-30)))
  print('==== %s: %d buffers' % (label, len(rows)))
  crossed = set()
  for i, (key, g, mx, rms) in enumerate(rows):
    flag = ''
    for th in (1e-3, 1e-2, 5e-2, 2e-1):
      if mx > th and (g, th) not in crossed:
        crossed.add((g, th))
        flag += ' <-- first %s > %g' % ('grad' if g else 'act', th)
    if flag or i % every == 0:
      print('%4d %-64s %s max %.4f rms %.5f%s' % (i, key[-64:], 'G' if g else 'A', mx, rms, flag))


def main():
  dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
  big = int(sys.argv[2]) if len(sys.argv) > 2 else 128
  s2 = tb._Step(dtype, 2)
  snap = {k: v.clone() for k, v in s2.eng._bufs.items() if v.dim() == 4 and v.shape[0] == 2}
  s2b = tb._Step(dtype, 2)
  compare(snap, s2b.eng._bufs, 1, '%s: two independent 2-image executors (noise floor)' % dtype, every=40)
  sN = tb._Step(dtype, big)
  compare(snap, sN.eng._bufs, big // 2, '%s: batch %d (tiled) vs batch 2' % (dtype, big))
  for k in ('cls_loss', 'box_loss', 'loss', 'gradient_norm'):
    print(k, s2.losses[k], sN.losses[k])


if __name__ == '__main__':
  main()
