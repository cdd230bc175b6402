"""Kernel lab: HIP-event timing of single C-ABI entry points at the layer shapes of EfficientDet-D0 640x640 batch 128
(bf16), with in-process A/B over the environment switches that the library reads per call.

The whole-step bench (bench.py) costs ~40 s of GPU time per variant; a kernel-level A/B here costs a few seconds:

  python scripts/kernel_lab.py --entry pw_bwd_weight --layers mid --ab EDET_PW_IMPL=auto,big,stream
  python scripts/kernel_lab.py --entry pw_bwd_data --shape 128x20x20x1152x192 --ab EDET_PW_BIG_MINKN=2048,1000000
  python scripts/kernel_lab.py --entry dw_bwd --layers all --reps 5
  python scripts/kernel_lab.py --list                                      # the layer tables, no GPU needed

Entries: pw_fwd, pw_bwd_data, pw_bwd_weight, pw_bwd (both gradients in one call) (1x1 convolutions; shape N x H x W x Cin x Cout), dw_fwd, dw_bwd
(depthwise; shape N x H x W x C x K x S).  Views are the ones the network uses: forward inputs carry BatchNorm + swish
(+ SE gate for the project layers) on load, gradients carry the BatchNorm backward on load and the statistic partials
in the epilogue.  Prints one line per (layer, variant): ms per call, algorithmic MB (input + output elements x 2 B,
the SURVEY 8d model), GB/s.  Wrap a variant run in `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... --` for counters
on exactly one kernel.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, H, Cin, Cout, kind) of the 1x1 convolutions of D0 at 640x640 (backbone expand / project, BiFPN, heads)
PW_LAYERS = [
    ('b0_project', 320, 32, 16, 'project'), ('b1_expand', 320, 16, 96, 'expand'),
    ('b1_project', 160, 96, 24, 'project'), ('b2_expand', 160, 24, 144, 'expand'),
    ('b2_project', 160, 144, 24, 'project'), ('b3_project', 80, 144, 40, 'project'),
    ('b4_expand', 80, 40, 240, 'expand'), ('b4_project', 80, 240, 40, 'project'),
    ('b5_project', 40, 240, 80, 'project'), ('b6_expand', 40, 80, 480, 'expand'),
    ('b6_project', 40, 480, 80, 'project'), ('b8_project', 40, 480, 112, 'project'),
    ('b9_expand', 40, 112, 672, 'expand'), ('b9_project', 40, 672, 112, 'project'),
    ('b11_project', 20, 672, 192, 'project'), ('b12_expand', 20, 192, 1152, 'expand'),
    ('b12_project', 20, 1152, 192, 'project'), ('b15_project', 20, 1152, 320, 'project'),
    ('fpn_80', 80, 64, 64, 'fpn'), ('fpn_40', 40, 64, 64, 'fpn'), ('fpn_20', 20, 64, 64, 'fpn'),
    ('fpn_10', 10, 64, 64, 'fpn'), ('fpn_5', 5, 64, 64, 'fpn'), ('cls_80', 80, 64, 810, 'fpn'),
    ('rs_80', 80, 40, 64, 'fpn'), ('rs_40', 40, 112, 64, 'fpn'), ('rs_20', 20, 320, 64, 'fpn'), ('box_80', 80, 64, 36, 'fpn'),
    ('b5_expand', 40, 40, 240, 'expand'), ('b3_expand', 80, 24, 144, 'expand'),
]
# (name, H_in, C, K, S) of the depthwise convolutions
DW_LAYERS = [
    ('b0', 320, 32, 3, 1), ('b1', 320, 96, 3, 2), ('b2', 160, 144, 3, 1), ('b3', 160, 144, 5, 2),
    ('b4', 80, 240, 5, 1), ('b5', 80, 240, 3, 2), ('b6', 40, 480, 3, 1), ('b8', 40, 480, 5, 1),
    ('b9', 40, 672, 5, 1), ('b11', 40, 672, 5, 2), ('b12', 20, 1152, 5, 1), ('b15', 20, 1152, 3, 1),
    ('fpn_80', 80, 64, 3, 1), ('fpn_40', 40, 64, 3, 1), ('fpn_20', 20, 64, 3, 1),
]
GROUPS = {'big': lambda h: h >= 160, 'mid': lambda h: 20 <= h <= 80, 'small': lambda h: h <= 10, 'all': lambda h: True}


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


def build_case(entry, shape):
  """-> (callable launching the entry once, algorithmic bytes)."""
  from automl_amd import _lib
  from automl_amd._lib import BwdEpi, call, ptr
  from tests import gpu_util as gu
  edt, tdt = _lib.EDET_BF16, torch.bfloat16
  dev = gu.DEV
  gen = torch.Generator(device=dev).manual_seed(0)

  def rand(*s):
    return torch.randn(s, device=dev, generator=gen).to(tdt)

  def vec(c, lo=0.5, hi=1.5):
    return (torch.rand(c, device=dev, generator=gen) * (hi - lo) + lo).float()
  npart = ctypes.c_int(0)
  parts = torch.zeros(_lib.MAX_PARTS * 2 * 4096, dtype=torch.float32, device=dev)
  wsp = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
  if entry.startswith('pw'):
    n, h, w, cin, cout = shape
    x = rand(n, h, w, gu.pad8(cin))
    gate = (torch.rand(n, cin, device=dev, generator=gen) * 0.8 + 0.2).float()
    tv = gu.tview(x, cin, vec(cin), vec(cin, -0.3, 0.3), gate if cin >= 96 and cin > cout else None, _lib.ACT_SWISH)
    if os.environ.get('EDET_LAB_PLAIN') == '1':      # a stored tensor as it is (BiFPN / tower layers)
      tv = gu.tview(x, cin)
    nbytes = n * h * w * (cin + cout) * 2
    if entry == 'pw_fwd':
      wt = rand(cout, gu.pad8(cin)) * (1.0 / np.sqrt(cin))
      out = torch.empty(n, h, w, gu.pad8(cout), dtype=tdt, device=dev)
      # EDET_LAB_NOSTATS=1 (read per call, so it can be the --ab variable): no BatchNorm statistic partials
      return (lambda: call('edet_pw_fwd', ctypes.byref(tv), ptr(wt), gu.pad8(cin), None, ptr(out), cout,
                           gu.pad8(cout), None if os.environ.get('EDET_LAB_NOSTATS') == '1' else ptr(parts),
                           ctypes.byref(npart), edt, gu.stream())), nbytes
    dz, y = rand(n, h, w, gu.pad8(cout)), rand(n, h, w, gu.pad8(cout))
    gv = gu.gview(dz, cout, y, vec(cout), vec(cout, -0.1, 0.1), vec(cout, -0.1, 0.1))
    if entry == 'pw_bwd_data':
      # the chain target: BatchNorm + swish view of the conv input, BatchNorm-backward sums in the epilogue
      tv = gu.tview(x, cin, vec(cin), vec(cin, -0.3, 0.3), None, _lib.ACT_SWISH)
      wk = rand(cin, gu.pad8(cout)) * (1.0 / np.sqrt(cout))
      gout = torch.empty(n, h, w, gu.pad8(cin), dtype=tdt, device=dev)
      mean, rstd = vec(cin, -0.2, 0.2), vec(cin)
      epi = BwdEpi(ptr(gout), 0, ptr(mean), ptr(rstd), ptr(parts), None)
      keep = (wk, gout, mean, rstd)
      return (lambda: (keep, call('edet_pw_bwd_data', ctypes.byref(gv), ptr(wk), gu.pad8(cout), ctypes.byref(tv),
                                  ctypes.byref(epi), ctypes.byref(npart), edt, gu.stream()))), nbytes
    if entry == 'pw_bwd':
      wk = rand(cin, gu.pad8(cout)) * (1.0 / np.sqrt(cout))
      gout = torch.empty(n, h, w, gu.pad8(cin), dtype=tdt, device=dev)
      mean, rstd = vec(cin, -0.2, 0.2), vec(cin)
      dwt = torch.zeros(cin, cout, dtype=torch.float32, device=dev)
      if cout in (64, 36) or (cin == cout and cin in (88, 112, 160, 224, 288, 384)):      # BiFPN / tower / resample layers (fpn_num_filters of d0 .. d7x): the input is a plain stored tensor, nothing to chain into
        tv = gu.tview(x, cin)
        if cout == 36:
          gv = gu.gview(dz, cout)      # predict layer: no BatchNorm behind it
        epi = BwdEpi(ptr(gout), 0, None, None, None, None)
        keep = (wk, gout, dwt)
      elif cin > cout:      # project layer: SE-gated view, the epilogue leaves D and the gate-gradient sums (engine._pw_bwd)
        dgate = torch.zeros(n, cin, dtype=torch.float32, device=dev)
        tv = gu.tview(x, cin, vec(cin), vec(cin, -0.3, 0.3), gate, _lib.ACT_SWISH)
        epi = BwdEpi(ptr(gout), 0, None, None, None, ptr(dgate))
        keep = (wk, gout, dgate, dwt)
      elif cout >= 2 * cin and cout != 810:
        # MBConv expansion: the input is a stored block output (plain view, nothing to chain into) and dy's saved
        # tensor is this convolution's own output -- the contract bit that lets the library leave y unread
        tv = gu.tview(x, cin)
        epi = BwdEpi(ptr(gout), 0, None, None, None, None, _lib.EPI_Y_IS_CONV_OF_INPUT)
        keep = (wk, gout, dwt)
      else:
        tv = gu.tview(x, cin, vec(cin), vec(cin, -0.3, 0.3), None, _lib.ACT_SWISH)
        epi = BwdEpi(ptr(gout), 0, ptr(mean), ptr(rstd), ptr(parts), None)
        keep = (wk, gout, mean, rstd, dwt)
      return (lambda: (keep, call('edet_pw_bwd', ctypes.byref(gv), ptr(wk), gu.pad8(cout), ctypes.byref(tv),
                                  ctypes.byref(epi), ctypes.byref(npart), ptr(dwt), ptr(wsp), wsp.numel() * 4, edt,
                                  gu.stream()))), 2 * nbytes
    if entry == 'pw_bwd_weight':
      dwt = torch.zeros(cin, cout, dtype=torch.float32, device=dev)
      return (lambda: call('edet_pw_bwd_weight', ctypes.byref(tv), ctypes.byref(gv), ptr(dwt), ptr(wsp),
                           wsp.numel() * 4, edt, gu.stream())), nbytes
  if entry.startswith('dw'):
    n, h, w, c, k, s = shape
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    x = rand(n, h, w, c)
    tv = gu.tview(x, c, vec(c), vec(c, -0.3, 0.3), None, _lib.ACT_SWISH)
    wk = (torch.randn(k, k, c, device=dev, generator=gen) / k).float()
    nbytes = n * (h * w + oh * ow) * c * 2
    if entry == 'dw_fwd':
      out = torch.empty(n, oh, ow, c, dtype=tdt, device=dev)
      return (lambda: call('edet_dw_fwd', ctypes.byref(tv), ptr(wk), k, s, ptr(out), c, ptr(parts),
                           ctypes.byref(npart), edt, gu.stream())), nbytes
    if entry == 'dw_bwd':
      dz, y = rand(n, oh, ow, c), rand(n, oh, ow, c)
      gv = gu.gview(dz, c, y, vec(c), vec(c, -0.1, 0.1), vec(c, -0.1, 0.1))
      gout = torch.empty(n, h, w, c, dtype=tdt, device=dev)
      mean, rstd = vec(c, -0.2, 0.2), vec(c)
      epi = BwdEpi(ptr(gout), 0, ptr(mean), ptr(rstd), ptr(parts), None)
      dwd = torch.zeros(k, k, c, dtype=torch.float32, device=dev)
      keep = (gout, mean, rstd)
      return (lambda: (keep, call('edet_dw_bwd', ctypes.byref(gv), ptr(wk), k, s, ctypes.byref(tv), ctypes.byref(epi),
                                  ctypes.byref(npart), ptr(dwd), ptr(wsp), wsp.numel() * 4, edt, gu.stream()))), nbytes
  raise SystemExit('unknown entry %s' % entry)


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument('--entry', default='pw_bwd_weight')
  ap.add_argument('--layers', default='', help='big | mid | small | all (by feature-map size), or a layer name')
  ap.add_argument('--shape', default='', help='NxHxWxCinxCout (pw) or NxHxWxCxKxS (dw)')
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--rounds', type=int, default=3, help='rotated measurement passes per layer (minimum reported)')
  ap.add_argument('--ab', default='', help='VAR=v1,v2,...: time every layer under each value of one environment switch')
  ap.add_argument('--list', action='store_true')
  ap.add_argument('--lib', default='', help='another build of the library to load instead of automl_amd/libedet_hip.so')
  args = ap.parse_args()
  if args.lib:
    from automl_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(args.lib)
  table = PW_LAYERS if args.entry.startswith('pw') else DW_LAYERS
  if args.list:
    for row in PW_LAYERS + DW_LAYERS:
      print(row)
    return
  cases = []
  if args.shape:
    cases.append(('custom', tuple(int(v) for v in args.shape.split('x'))))
  else:
    sel = GROUPS.get(args.layers or 'all')
    for row in table:
      if (sel and sel(row[1])) or (not sel and row[0] == args.layers):
        shape = (args.batch, row[1], row[1], row[2], row[3]) if args.entry.startswith('pw') else \
            (args.batch, row[1], row[1], row[2], row[3], row[4])
        cases.append((row[0], shape))
  var, values = None, [None]
  if args.ab:
    var, vals = args.ab.split('=', 1)
    values = vals.split(',')
  total = {v: 0.0 for v in values}

  def setenv(v):
    if var:
      if v in ('', 'unset'):
        os.environ.pop(var, None)
      else:
        os.environ[var] = v
  for name, shape in cases:
    fn, nbytes = build_case(args.entry, shape)
    # r03e: the variant measured FIRST after a case is built came out up to 15 % slow (the same code path measured 0.507
    # first and 0.450 third).  Every variant is therefore warmed once, then measured in `rounds` rotated passes; the
    # minimum over the passes is reported.
    best = {v: float('inf') for v in values}
    for v in values:
      setenv(v)
      timed(fn, 2)
    for rnd in range(args.rounds):
      order = values[rnd % len(values):] + values[:rnd % len(values)]
      for v in order:
        setenv(v)
        best[v] = min(best[v], timed(fn, args.reps))
    for v in values:
      ms = best[v]
      total[v] += ms
      print('%-14s %-26s %-22s %9.4f ms %9.1f MB %8.1f GB/s' % (
          args.entry, '%s %s' % (name, 'x'.join(map(str, shape[1:]))), '%s=%s' % (var, v) if var else '', ms,
          nbytes / 1e6, nbytes / ms / 1e6))
    del fn
    torch.cuda.empty_cache()
  for v in values:
    print('TOTAL %-22s %9.4f ms over %d layers' % ('%s=%s' % (var, v) if var else '', total[v], len(cases)))


if __name__ == '__main__':
  main()
