"""ctypes binding of the network-level C ABI (include/edet_net.h): a recorded step plan replayed by the library's own
host runtime (csrc/net_runtime.cpp).  What a compiled host would call; here for the parity tests and for Python callers
that want a step without the engine (no tape, no per-launch Python)."""
import ctypes

import numpy as np

from automl_amd import _lib

ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

NET_SIGNATURES = {
    'edet_create': [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)],
    'edet_destroy': [ctypes.c_void_p],
    'edet_net_buffer': [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)],
    'edet_net_num_buffers': [ctypes.c_void_p],
    'edet_net_property': [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)],
    'edet_net_has_program': [ctypes.c_void_p, ctypes.c_char_p],
    'edet_net_use_graph': [ctypes.c_void_p, ctypes.c_int],
    'edet_copy_to_host': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t],
    'edet_copy_to_device': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t],
    'edet_forward': [ctypes.c_void_p, ctypes.c_void_p],
    'edet_detect': [ctypes.c_void_p, ctypes.c_void_p],
    'edet_train_step': [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p],
    'edet_dp_init': [ctypes.c_void_p, ALLREDUCE_FN, ctypes.c_void_p],
    'edet_anchors': [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double,
                     ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)],
}
_bound = False


def _lib_net():
  global _bound
  lib = _lib.load()
  if not _bound:
    for name, argtypes in NET_SIGNATURES.items():
      fn = getattr(lib, name)
      fn.restype = ctypes.c_int
      fn.argtypes = argtypes
    lib.edet_net_buffer_name.restype = ctypes.c_char_p
    lib.edet_net_buffer_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _bound = True
  return lib


def _check(lib, rc, what):
  if rc != 0:
    raise _lib.EdetError('%s failed (%d): %s' % (what, rc, lib.edet_last_error().decode()))


def copy_to_host(device_ptr, nbytes):
  """uint8 numpy copy of nbytes of device memory (waits for the device)."""
  lib = _lib_net()
  out = np.empty(nbytes, np.uint8)
  _check(lib, lib.edet_copy_to_host(out.ctypes.data, device_ptr, nbytes), 'edet_copy_to_host')
  return out


def anchors(min_level, max_level, num_scales, aspect_ratios, anchor_scale, image_size):
  """tf2/anchors.py Anchors(...).boxes through edet_anchors: float32 [N, 4] (host only, no GPU needed)."""
  lib = _lib_net()
  h, w = (image_size, image_size) if isinstance(image_size, int) else image_size
  ar = (ctypes.c_double * len(aspect_ratios))(*[float(a) for a in aspect_ratios])
  count = ctypes.c_int64(0)
  _check(lib, lib.edet_anchors(min_level, max_level, num_scales, ar, len(aspect_ratios), float(anchor_scale), h, w, None, 0,
                               ctypes.byref(count)), 'edet_anchors')
  out = np.empty((count.value, 4), np.float32)
  _check(lib, lib.edet_anchors(min_level, max_level, num_scales, ar, len(aspect_ratios), float(anchor_scale), h, w,
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), count.value, ctypes.byref(count)),
         'edet_anchors')
  return out


class CNet(object):
  """A loaded plan: forward() / train_step() run entirely inside the library."""

  def __init__(self, plan_path):
    self.lib = _lib_net()
    h = ctypes.c_void_p()
    _check(self.lib, self.lib.edet_create(plan_path.encode(), ctypes.byref(h)), 'edet_create')
    self.h = h
    self._cb = None

  def close(self):
    if self.h:
      self.lib.edet_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:      # noqa: BLE001 -- interpreter shutdown
      pass

  def names(self):
    return [self.lib.edet_net_buffer_name(self.h, i).decode() for i in range(self.lib.edet_net_num_buffers(self.h))]

  def buffer(self, name):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    _check(self.lib, self.lib.edet_net_buffer(self.h, name.encode(), ctypes.byref(p), ctypes.byref(n)), 'edet_net_buffer')
    return p.value, n.value

  def prop(self, name):
    v = ctypes.c_int64()
    _check(self.lib, self.lib.edet_net_property(self.h, name.encode(), ctypes.byref(v)), 'edet_net_property')
    return v.value

  def has_program(self, name):
    return self.lib.edet_net_has_program(self.h, name.encode()) == 1

  def use_graph(self, on=True):
    _check(self.lib, self.lib.edet_net_use_graph(self.h, 1 if on else 0), 'edet_net_use_graph')

  def forward(self, stream=None):
    _check(self.lib, self.lib.edet_forward(self.h, stream), 'edet_forward')

  def detect(self, stream=None):
    _check(self.lib, self.lib.edet_detect(self.h, stream), 'edet_detect')

  def train_step(self, learning_rate, ema_decay=0.0, stream=None):
    _check(self.lib, self.lib.edet_train_step(self.h, learning_rate, ema_decay, stream), 'edet_train_step')

  def dp_init(self, fn):
    """fn(buf_ptr, count, stream) -> 0; None = single replica."""
    if fn is None:
      self._cb = ctypes.cast(None, ALLREDUCE_FN)
    else:
      self._cb = ALLREDUCE_FN(lambda ctx, buf, count, stream: int(fn(buf, count, stream) or 0))
    _check(self.lib, self.lib.edet_dp_init(self.h, self._cb, None), 'edet_dp_init')

  def read(self, name):
    """Bytes of a named buffer (synchronises the device)."""
    p, n = self.buffer(name)
    out = np.empty(n, np.uint8)
    _check(self.lib, self.lib.edet_copy_to_host(out.ctypes.data, p, n), 'edet_copy_to_host')
    return out

  def write(self, name, array):
    a = np.ascontiguousarray(array)
    p, n = self.buffer(name)
    assert a.nbytes == n, (name, a.nbytes, n)
    _check(self.lib, self.lib.edet_copy_to_device(p, a.ctypes.data, n), 'edet_copy_to_device')
