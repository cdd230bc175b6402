"""Teacher forcing for the bf16-storage-emulating oracle -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/ and the checker legs of bench.py (parity block) use this.  See Oracle.hook in efficientdet_oracle.py.
"""
import torch


class TeacherForce(object):
  """Hook for oracle.Oracle(storage='bf16').hook: compares every value / gradient the oracle is about to store with
  the buffer the device executor stored under the same key, records max |difference| / max |oracle value|, and hands
  the DEVICE tensor back, so that the next layer of the oracle starts from the device's own data (layer-local
  parity: no end-to-end amplification).  Buffers are NHWC with a padded channel stride; the oracle works in NCHW."""

  def __init__(self, eng, grad_scale=1.0):
    self.bufs = eng._bufs
    self.fwd_err, self.bwd_err, self.missing = {}, {}, []
    self.grad_scale = grad_scale

  def _device(self, key, like):
    t = self.bufs.get(key)
    if t is None or t.dim() != 4:
      return None
    c = like.shape[1]
    d = t[..., :c].permute(0, 3, 1, 2).to('cpu', torch.float32)
    assert tuple(d.shape) == tuple(like.shape), (key, tuple(d.shape), tuple(like.shape))
    return d

  @staticmethod
  def _err(dev, ref):
    return float((dev - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)

  def fwd(self, key, value):
    d = self._device(key, value)
    if d is None:
      self.missing.append(key)
      return value
    self.fwd_err[key] = self._err(d, value)
    return d

  def bwd(self, key, grad):
    d = self._device(key, grad)
    if d is None:
      self.missing.append(key)
      return grad
    if self.grad_scale != 1.0:
      d = d * self.grad_scale
    self.bwd_err[key] = self._err(d, grad)
    return d

  @staticmethod
  def worst(errs, k=3):
    return sorted(((round(v, 5), key) for key, v in errs.items()), reverse=True)[:k]
