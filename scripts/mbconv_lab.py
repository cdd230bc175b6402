"""Lab: the fused MBConv head (edet_mbconv_expand_stats + edet_mbconv_expand_dw_fwd) against the two-kernel path
(edet_pw_fwd + edet_dw_fwd) at the three layer shapes of EfficientDet-D0 640x640 batch 128 that the engine fuses.
HIP-event time per call; environment switches of the library (EDET_MBF_*) are read per call.

  python scripts/mbconv_lab.py [--batch 128] [--reps 10]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import _lib  # noqa: E402
from automl_amd._lib import ACT_NONE, ACT_SWISH, TView, call, ptr  # noqa: E402

LAYERS = [('b1', 320, 16, 96, 3, 2, True), ('b2', 160, 24, 144, 3, 1, True), ('b3', 160, 24, 144, 5, 2, False)]


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
  ap.add_argument('--only', default='')
  ap.add_argument('--custom', default='', help='HWxCINxCEXPxKxS: one more layer shape')
  args = ap.parse_args()
  dev = 'cuda:0'
  st = torch.cuda.current_stream().cuda_stream
  n = args.batch
  bf = _lib.EDET_BF16
  layers = list(LAYERS)
  if args.custom:
    hw_, cin_, cexp_, k_, s_ = (int(v) for v in args.custom.split('x'))
    layers = [('custom', hw_, cin_, cexp_, k_, s_, True)]
  for name, hw, cin, cexp, k, s, affine in layers:
    if args.only and name not in args.only.split(','):
      continue
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((n, hw, hw, cin), generator=g, device=dev).to(torch.bfloat16)
    wk = (torch.randn((cin, cexp), generator=g, device=dev) / np.sqrt(cin)).float()
    wt = torch.zeros(cexp, cin, dtype=torch.bfloat16, device=dev)
    call('edet_cast_matrix', ptr(wk), ptr(wt), cin, cexp, cin, 1, bf, st)
    isc = (1 + 0.1 * torch.randn(cin, generator=g, device=dev)).float() if affine else None
    ish = (0.1 * torch.randn(cin, generator=g, device=dev)).float() if affine else None
    esc = (1 + 0.1 * torch.randn(cexp, generator=g, device=dev)).float()
    esh = (0.1 * torch.randn(cexp, generator=g, device=dev)).float()
    dww = (torch.randn((k, k, cexp), generator=g, device=dev) / k).float()
    oh = (hw + s - 1) // s
    e = torch.empty((n, hw, hw, cexp), dtype=torch.bfloat16, device=dev)
    out = torch.empty((n, oh, oh, cexp), dtype=torch.bfloat16, device=dev)
    parts = torch.zeros(_lib.MAX_PARTS * 2 * cexp, dtype=torch.float32, device=dev)
    npart = ctypes.c_int(0)
    tv = TView(ptr(x), ptr(isc), ptr(ish), None, ACT_NONE, n, hw, hw, cin, cin)
    tve = TView(ptr(e), ptr(esc), ptr(esh), None, ACT_SWISH, n, hw, hw, cexp, cexp)

    def stats():
      call('edet_mbconv_expand_stats', ctypes.byref(tv), ptr(wt), cin, cexp, ptr(parts), ctypes.byref(npart), bf, st)

    def fused(train):
      call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), cin, cexp, ptr(esc), ptr(esh), ACT_SWISH,
           ptr(e) if train else None, cexp, ptr(dww), k, s, ptr(out), cexp, ptr(parts) if train else None,
           ctypes.byref(npart), bf, st)

    def pw():
      call('edet_pw_fwd', ctypes.byref(tv), ptr(wt), cin, None, ptr(e), cexp, cexp, ptr(parts), ctypes.byref(npart), bf, st)

    def dw():
      call('edet_dw_fwd', ctypes.byref(tve), ptr(dww), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart), bf, st)

    fill_ms = timed(lambda: e.zero_(), args.reps)
    print('   (fill of the expanded tensor, %.2f GB: %.3f ms = %.2f TB/s written)' % (e.numel() * 2 / 1e9, fill_ms, e.numel() * 2 / fill_ms / 1e9))
    t = {'stats': timed(stats, args.reps), 'fused_train': timed(lambda: fused(True), args.reps),
         'fused_infer': timed(lambda: fused(False), args.reps), 'pw_fwd': timed(pw, args.reps), 'dw_fwd': timed(dw, args.reps)}
    print('%s %dx%dx%d->%d k%ds%d  stats %.3f  fused_train %.3f  (sum %.3f)  fused_infer %.3f | pw_fwd %.3f  dw_fwd %.3f  (sum %.3f)' % (
        name, hw, hw, cin, cexp, k, s, t['stats'], t['fused_train'], t['stats'] + t['fused_train'], t['fused_infer'],
        t['pw_fwd'], t['dw_fwd'], t['pw_fwd'] + t['dw_fwd']), flush=True)


if __name__ == '__main__':
  main()
