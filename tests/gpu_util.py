"""Helpers for the -m gpu parity tests: torch tensors <-> C-ABI views, comparison with tolerances."""
import ctypes
import zlib

import numpy as np
import torch

from automl_amd import _lib
from automl_amd._lib import BwdEpi, GView, TView, call, ptr

DEV = 'cuda:0'
DTYPES = [('f32', _lib.EDET_F32, torch.float32), ('bf16', _lib.EDET_BF16, torch.bfloat16)]


def stream():
  return torch.cuda.current_stream().cuda_stream


def pad8(c):
  return (c + 7) // 8 * 8


def rnd(rng, shape, tdt, scale=1.0):
  """Random fp32 CPU tensor already rounded to the storage dtype (so the oracle sees the same values)."""
  t = torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))
  return t.to(tdt).to(torch.float32)


def to_dev(t, tdt, ld=None):
  """CPU fp32 [n,h,w,c] -> device tensor of dtype tdt with channel stride ld (zero padded)."""
  c = t.shape[-1]
  ld = ld or pad8(c)
  out = torch.zeros(t.shape[:-1] + (ld,), dtype=tdt, device=DEV)
  out[..., :c] = t.to(device=DEV, dtype=tdt)
  return out


def fdev(t):
  return None if t is None else t.to(device=DEV, dtype=torch.float32).contiguous()


def _dev_f32(t):
  if t is None:
    return None
  if t.device.type != 'cuda':
    t = fdev(t)
  return t


def tview(data, c, scale=None, shift=None, gate=None, act=0):
  """edet_tview_t over device tensors; fp32 vectors may be given on the CPU (moved and kept alive)."""
  n, h, w, ld = data.shape
  keep = [data] + [_dev_f32(t) for t in (scale, shift, gate)]
  tv = TView(ptr(data), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), act, n, h, w, c, ld)
  tv._keep = keep      # the struct only holds raw pointers
  return tv


def gview(dz, c, y=None, a=None, b=None, cc=None):
  n, h, w, ld = dz.shape
  keep = [dz, y] + [_dev_f32(t) for t in (a, b, cc)]
  gv = GView(ptr(dz), ptr(y), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), n, h, w, c, ld)
  gv._keep = keep
  return gv


def tol_for(name):
  # fp32 path: accumulation-order differences only.  bf16 path: one storage rounding (2^-8 rel)
  # of the output plus bf16 operand products accumulated in fp32.
  return (2e-4, 2e-5) if name == 'f32' else (2e-2, 2e-2)


def check(got, want, name, what, rtol=None, atol=None, scale_by_max=True):
  got = got.detach().to('cpu', torch.float32)
  want = want.detach().to('cpu', torch.float32)
  assert got.shape == want.shape, (what, got.shape, want.shape)
  r, a = tol_for(name)
  rtol = r if rtol is None else rtol
  atol = a if atol is None else atol
  ref = float(want.abs().max()) if scale_by_max else 1.0
  err = float((got - want).abs().max())
  bound = rtol * ref + atol * (1.0 if not scale_by_max else min(1.0, max(ref, 1e-30)))
  assert np.isfinite(err) and err <= bound, '%s [%s]: max abs err %.3e > %.3e (max |ref| %.3e)' % (
      what, name, err, bound, ref)
  return err


def sum_partials(partials, nparts, c):
  p = partials[:nparts * 2 * c].view(nparts, 2, c).double().sum(0)
  return p[0].float().cpu(), p[1].float().cpu()


def seed_of(*args):
  """Deterministic per-case seed (python's hash() is salted per process)."""
  return zlib.crc32(repr(args).encode()) & 0x7fffffff


from oracle.teacher_force import TeacherForce  # noqa: E402,F401  (re-exported for the tests)
