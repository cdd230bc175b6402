"""-m gpu: the fused MBConv head (expansion 1x1 -> BatchNorm -> activation -> depthwise k x k in one kernel,
automl_amd/csrc/mbconv_fused.hip) through the C ABI against the CPU oracle's ops.

Reference: efficientdet/backbone/efficientnet_model.py:378-392 (x = act(bn0(expand_conv(x))); x = act(bn1(dw(x)))).
"""
import ctypes

import numpy as np
import pytest
import torch

from automl_amd import _lib
from automl_amd._lib import ACT_HSWISH, ACT_NONE, ACT_RELU6, ACT_SWISH, call, ptr
from oracle import efficientdet_oracle as orc
from tests import gpu_util as gu
from tests.test_gpu_kernels import act_oracle, dw_oracle, partial_buf

pytestmark = pytest.mark.gpu

NP = ctypes.c_int
BF = ('bf16', _lib.EDET_BF16, torch.bfloat16)

# (n, h, w, cin, cexp): ragged maps, several column windows (w > 62 / 31 outputs), several row tiles (oh > 40), the
# channel counts of the layers the engine fuses (16 -> 96, 24 -> 144) and the envelope's corners (8, 32 input channels)
SHAPES = [(2, 9, 11, 16, 96), (1, 70, 67, 24, 144), (2, 33, 130, 16, 48), (1, 20, 20, 32, 144), (3, 5, 5, 8, 48),
          (1, 170, 9, 24, 144), (2, 64, 64, 16, 96), (1, 63, 125, 24, 96), (2, 31, 40, 32, 96)]
KS = [(3, 1), (3, 2), (5, 2)]       # the (kernel, stride) pairs of the MBConv stages with <= 32 block-input channels


def _bf(t):
  return t.to(torch.bfloat16).float()


def _problem(shape, ks, affine, act):
  n, h, w, cin, cexp = shape
  k, s = ks
  rng = np.random.default_rng(gu.seed_of((shape, ks, affine, act)))
  x = gu.rnd(rng, (n, h, w, cin), torch.bfloat16)
  isc = ish = None
  if affine:
    isc = torch.from_numpy((1 + 0.3 * rng.standard_normal(cin)).astype(np.float32))
    ish = torch.from_numpy((0.3 * rng.standard_normal(cin)).astype(np.float32))
  wk = gu.rnd(rng, (cin, cexp), torch.bfloat16, 1.0 / np.sqrt(cin))
  esc = torch.from_numpy((1 + 0.3 * rng.standard_normal(cexp)).astype(np.float32))
  esh = torch.from_numpy((0.3 * rng.standard_normal(cexp)).astype(np.float32))
  dww = torch.from_numpy((rng.standard_normal((k, k, cexp)) / k).astype(np.float32))
  # oracle: the expansion as the matrix cores see it (bf16 operands, fp32 accumulation), stored as bf16
  xt = x * isc + ish if affine else x
  e = (_bf(xt).reshape(-1, cin) @ wk).reshape(n, h, w, cexp)
  return x, isc, ish, wk, esc, esh, dww, e


def _device_inputs(x, isc, ish, wk, cin, cexp):
  xd = gu.to_dev(x, torch.bfloat16)
  ldk = gu.pad8(cin)
  wt = torch.zeros(cexp, ldk, dtype=torch.bfloat16, device=gu.DEV)
  wkd = gu.fdev(wk)
  call('edet_cast_matrix', ptr(wkd), ptr(wt), cin, cexp, ldk, 1, _lib.EDET_BF16, gu.stream())
  tv = gu.tview(xd, cin, isc, ish, None, ACT_NONE)
  return xd, wt, ldk, tv


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('affine', [False, True])
def test_expand_stats(shape, affine):
  n, h, w, cin, cexp = shape
  x, isc, ish, wk, _, _, _, e = _problem(shape, (3, 1), affine, ACT_SWISH)
  xd, wt, ldk, tv = _device_inputs(x, isc, ish, wk, cin, cexp)
  assert _lib.load().edet_mbconv_fused_supported(ctypes.byref(tv), cexp, 3, 1, _lib.EDET_BF16) == 1
  parts = partial_buf(cexp)
  npart = NP(0)
  call('edet_mbconv_expand_stats', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(parts), ctypes.byref(npart),
       _lib.EDET_BF16, gu.stream())
  torch.cuda.synchronize()
  s1, s2 = gu.sum_partials(parts, npart.value, cexp)
  er = _bf(e)
  rows = n * h * w
  # sums of the bf16-ROUNDED products: a rounding flip moves one element by 2^-8 of its value
  gu.check(s1, er.sum((0, 1, 2)), 'bf16', 'expand stats sum', rtol=2e-3, atol=2e-3 * rows, scale_by_max=False)
  gu.check(s2, (er * er).sum((0, 1, 2)), 'bf16', 'expand stats sumsq', rtol=2e-3)
  # and bit-identical from run to run (fixed summation tree, no atomics)
  parts2 = partial_buf(cexp)
  call('edet_mbconv_expand_stats', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(parts2), ctypes.byref(npart),
       _lib.EDET_BF16, gu.stream())
  torch.cuda.synchronize()
  assert torch.equal(parts[:npart.value * 2 * cexp], parts2[:npart.value * 2 * cexp])


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('ks', KS)
@pytest.mark.parametrize('mode', ['train_affine', 'train_plain', 'infer_affine'])
def test_expand_dw_fwd(shape, ks, mode, act=ACT_SWISH):
  n, h, w, cin, cexp = shape
  k, s = ks
  affine = mode.endswith('affine')
  train = mode.startswith('train')
  x, isc, ish, wk, esc, esh, dww, e = _problem(shape, ks, affine, act)
  er = _bf(e)
  xd, wt, ldk, tv = _device_inputs(x, isc, ish, wk, cin, cexp)
  oh, ow = (h + s - 1) // s, (w + s - 1) // s
  out = torch.full((n, oh, ow, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV)
  eout = torch.full((n, h, w, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV) if train else None
  parts = partial_buf(cexp)
  npart = NP(0)
  escd, eshd, dwwd = gu.fdev(esc), gu.fdev(esh), gu.fdev(dww)      # (kept alive: the call takes raw pointers)
  call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(escd), ptr(eshd), act,
       ptr(eout), cexp, ptr(dwwd), k, s, ptr(out), cexp, ptr(parts) if train else None,
       ctypes.byref(npart), _lib.EDET_BF16, gu.stream())
  torch.cuda.synchronize()
  what = 'mbconv fused %s k%d s%d %s' % (shape, k, s, mode)
  if train:
    # the stored expanded tensor: every element written exactly once, equal to the oracle's product up to a rounding flip
    assert bool(torch.isfinite(eout.float()).all()), what + ': expanded tensor has unwritten elements'
    gu.check(eout, er, 'bf16', what + ' expanded')
    er = eout.float().cpu()          # the depthwise half is checked against what the device itself stored
  z = er * esc + esh
  want = dw_oracle(act_oracle(z, act), dww, k, s)
  assert bool(torch.isfinite(out.float()).all()), what + ': output has unwritten elements'
  gu.check(out, want, 'bf16', what)
  if train:
    s1, s2 = gu.sum_partials(parts, npart.value, cexp)
    of = out.float().cpu()
    gu.check(s1, of.sum((0, 1, 2)), 'bf16', what + ' sum vs stored', rtol=1e-3, atol=1e-3 * n * oh * ow, scale_by_max=False)
    gu.check(s2, (of * of).sum((0, 1, 2)), 'bf16', what + ' sumsq vs stored', rtol=1e-3)


@pytest.mark.parametrize('act', [ACT_RELU6, ACT_HSWISH])
def test_expand_dw_fwd_other_activations(act):
  """relu6 / hswish (the lite models, utils.activation_fn utils.py:36-53): the ACTM = 2 instantiations."""
  for ks in ((3, 2), (5, 2)):
    test_expand_dw_fwd((2, 33, 70, 16, 96), ks, 'train_affine', act)


def test_layers_outside_the_envelope_are_refused():
  """edet_mbconv_fused_supported says 0 and the entry points fail with a message (the engine then runs edet_pw_fwd +
  edet_dw_fwd): more than 32 block-input channels, an expansion width that is not 1..3 groups of 48, 5x5 stride 1."""
  lib = _lib.load()
  for (cin, cexp, k, s) in ((40, 240, 3, 1), (32, 192, 3, 1), (16, 64, 3, 2), (24, 144, 5, 1)):
    xd = torch.zeros((1, 8, 8, cin), dtype=torch.bfloat16, device=gu.DEV)
    tv = gu.tview(xd, cin, None, None, None, ACT_NONE)
    assert lib.edet_mbconv_fused_supported(ctypes.byref(tv), cexp, k, s, _lib.EDET_BF16) == 0, (cin, cexp, k, s)
    wt = torch.zeros(cexp, cin, dtype=torch.bfloat16, device=gu.DEV)
    sc = torch.ones(cexp, dtype=torch.float32, device=gu.DEV)
    dww = torch.zeros((k, k, cexp), dtype=torch.float32, device=gu.DEV)
    out = torch.zeros((1, 8, 8, cexp), dtype=torch.bfloat16, device=gu.DEV)
    with pytest.raises(_lib.EdetError, match='unsupported layer'):
      call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), cin, cexp, ptr(sc), ptr(sc), ACT_SWISH, None, 0,
           ptr(dww), k, s, ptr(out), cexp, None, None, _lib.EDET_BF16, gu.stream())
  xd = torch.zeros((1, 8, 8, 16), dtype=torch.float32, device=gu.DEV)
  tv = gu.tview(xd, 16, None, None, None, ACT_NONE)
  assert lib.edet_mbconv_fused_supported(ctypes.byref(tv), 96, 3, 2, _lib.EDET_F32) == 0      # fp32 storage: the two-kernel path


def test_fused_equals_the_two_kernel_path_bit_for_bit_in_storage():
  """edet_pw_fwd + edet_dw_fwd on the same layer: the stored expanded tensors agree up to rounding flips of the two MFMA
  shapes' summation orders (<= 1 bf16 ulp on a handful of elements), the depthwise outputs to the bf16 tolerance."""
  shape, (k, s) = (2, 40, 70, 24, 144), (5, 2)
  n, h, w, cin, cexp = shape
  x, isc, ish, wk, esc, esh, dww, e = _problem(shape, (k, s), True, ACT_SWISH)
  xd, wt, ldk, tv = _device_inputs(x, isc, ish, wk, cin, cexp)
  oh, ow = (h + s - 1) // s, (w + s - 1) // s
  outs = []
  escd, eshd, dwwd = gu.fdev(esc), gu.fdev(esh), gu.fdev(dww)
  for fused in (True, False):
    out = torch.zeros((n, oh, ow, cexp), dtype=torch.bfloat16, device=gu.DEV)
    eout = torch.zeros((n, h, w, cexp), dtype=torch.bfloat16, device=gu.DEV)
    parts, npart = partial_buf(cexp), NP(0)
    if fused:
      call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(escd), ptr(eshd),
           ACT_SWISH, ptr(eout), cexp, ptr(dwwd), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart),
           _lib.EDET_BF16, gu.stream())
    else:
      call('edet_pw_fwd', ctypes.byref(tv), ptr(wt), ldk, None, ptr(eout), cexp, cexp, ptr(parts), ctypes.byref(npart),
           _lib.EDET_BF16, gu.stream())
      tv2 = gu.tview(eout, cexp, esc, esh, None, ACT_SWISH)
      call('edet_dw_fwd', ctypes.byref(tv2), ptr(dwwd), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart),
           _lib.EDET_BF16, gu.stream())
    torch.cuda.synchronize()
    outs.append((eout.float().cpu(), out.float().cpu()))
  (e0, o0), (e1, o1) = outs
  flips = int((e0 != e1).sum())
  assert flips <= 1e-3 * e0.numel(), flips
  assert float((e0 - e1).abs().max()) <= 2.0 ** -7 * float(e0.abs().max())
  gu.check(o0, o1, 'bf16', 'fused vs two-kernel depthwise output')


@pytest.mark.parametrize('layer', [(320, 16, 96, 3, 2), (160, 24, 144, 3, 1), (160, 24, 144, 5, 2)],
                         ids=lambda l: '%dx%dx%d->%d_k%ds%d' % (l[0], l[0], l[1], l[2], l[3], l[4]))
def test_batch_128_equals_64_copies_of_the_two_image_run(layer):
  """The three layers the engine fuses at EfficientDet-D0 640x640 batch 128 (several tiles per workgroup, every XCD
  walking its own tile range): 64 copies of two images must give 64 copies of the two-image result BIT FOR BIT --
  the stored expanded tensor and the depthwise output.  (r06: with the expanded row leaving through buffer_store_dwordx4
  the first dword of some chunks was wrong at this size only; the two-image result itself is oracle-checked above.)"""
  hw, cin, cexp, k, s = layer
  n = 128
  rng = np.random.default_rng(gu.seed_of(layer))
  x2 = gu.rnd(rng, (2, hw, hw, cin), torch.bfloat16)
  wk = gu.rnd(rng, (cin, cexp), torch.bfloat16, 1.0 / np.sqrt(cin))
  isc = torch.from_numpy((1 + 0.3 * rng.standard_normal(cin)).astype(np.float32))
  ish = torch.from_numpy((0.3 * rng.standard_normal(cin)).astype(np.float32))
  escd = gu.fdev(torch.from_numpy((1 + 0.3 * rng.standard_normal(cexp)).astype(np.float32)))
  eshd = gu.fdev(torch.from_numpy((0.3 * rng.standard_normal(cexp)).astype(np.float32)))
  dwwd = gu.fdev(torch.from_numpy((rng.standard_normal((k, k, cexp)) / k).astype(np.float32)))
  oh = (hw + s - 1) // s
  res = {}
  for nn in (2, n):
    x = x2 if nn == 2 else x2.repeat(nn // 2, 1, 1, 1)
    xd, wt, ldk, tv = _device_inputs(x, isc, ish, wk, cin, cexp)
    out = torch.full((nn, oh, oh, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV)
    eout = torch.full((nn, hw, hw, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV)
    parts, npart = partial_buf(cexp), NP(0)
    call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(escd), ptr(eshd), ACT_SWISH,
         ptr(eout), cexp, ptr(dwwd), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart), _lib.EDET_BF16, gu.stream())
    torch.cuda.synchronize()
    res[nn] = (eout, out, gu.sum_partials(parts, npart.value, cexp))
  e2, o2, (s1_2, s2_2) = res[2]
  e, o, (s1, s2) = res[n]
  assert bool(torch.isfinite(e2.float()).all()) and bool(torch.isfinite(o2.float()).all())
  e = e.view((n // 2, 2) + tuple(e2.shape[1:]))
  o = o.view((n // 2, 2) + tuple(o2.shape[1:]))
  for kk in range(n // 2):
    assert torch.equal(e[kk].view(torch.int16), e2.view(torch.int16)), ('expanded tensor, image pair', kk)
    assert torch.equal(o[kk].view(torch.int16), o2.view(torch.int16)), ('depthwise output, image pair', kk)
  gu.check(s1, s1_2 * (n // 2), 'bf16', 'statistics sum', rtol=1e-4, atol=1e-5 * n * oh * oh, scale_by_max=False)
  gu.check(s2, s2_2 * (n // 2), 'bf16', 'statistics sumsq', rtol=1e-4)
