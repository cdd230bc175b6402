"""HIP-event timing of edet_conv_fwd (dense k x k convolution, bf16, inference view) at given shapes, with in-process A/B over
environment switches the library reads per call.
  python scripts/bench_conv.py --shapes 256x112x112x24x24,256x56x56x48x192 --ab EDET_CONV_HALO=0,1
Shape: N x H x W x Cin x Cout (3 x 3, stride 1).  Prints ms per call, algorithmic MB (input + output x 2 B) and GB/s."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--shapes', required=True)
  ap.add_argument('--ab', default='')
  ap.add_argument('--env', default='', help='VAR=val,... set for the whole run')
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--stride', type=int, default=1)
  args = ap.parse_args()
  for kv in filter(None, args.env.split(',')):
    k, v = kv.split('=')
    os.environ[k] = v
  from automl_amd import _lib
  from automl_amd._lib import call, ptr
  from tests import gpu_util as gu
  var, vals = (args.ab.split('=')[0], args.ab.split('=')[1].split(',')) if args.ab else ('', [''])
  edt, tdt = _lib.EDET_BF16, torch.bfloat16
  for sh in args.shapes.split(','):
    n, h, w, cin, cout = [int(x) for x in sh.split('x')]
    x = torch.randn(n, h, w, cin, device=gu.DEV).to(tdt)
    wt = (torch.randn(cout, 9 * cin, device=gu.DEV) / (9 * cin) ** 0.5).to(tdt)
    oh, ow = (h + args.stride - 1) // args.stride, (w + args.stride - 1) // args.stride
    out = torch.empty(n, oh, ow, gu.pad8(cout), dtype=tdt, device=gu.DEV)
    sc = torch.rand(cin, device=gu.DEV) + 0.5
    shf = torch.rand(cin, device=gu.DEV) - 0.5
    tv = gu.tview(x, cin, sc, shf, None, _lib.ACT_NONE)
    npart = ctypes.c_int(0)
    for v in vals:
      if var:
        os.environ[var] = v
      fn = lambda: call('edet_conv_fwd', ctypes.byref(tv), ptr(wt), 9 * cin, 3, args.stride, ptr(out), cout, gu.pad8(cout), None,
                        ctypes.byref(npart), edt, gu.stream())
      for _ in range(3):
        fn()
      torch.cuda.synchronize()
      best = 1e9
      for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
          fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.reps)
      mb = (n * h * w * cin + n * oh * ow * cout) * 2 / 1e6
      print('conv3x3 s%d %-22s %s=%-4s %8.4f ms %8.1f MB %8.1f GB/s' % (args.stride, sh, var, v, best, mb, mb / best))


if __name__ == '__main__':
  main()
