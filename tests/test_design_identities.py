"""Algebra behind kernels that are planned but not built yet (plan for round 5), checked in float64 on the
CPU so that the plan rests on something executable."""
import numpy as np
import torch


def test_se_backward_sums_factor_into_per_image_tables():
  """MBConv tail (efficientnet_model.py:153-195,386-410): y = depthwise output, z = BatchNorm(y), h = swish(z), gate g[n,c] from
  the pooled h, u = h * g -> projection.  The BatchNorm backward of y needs S1 = sum dz and S2 = sum dz * zhat over (n, h, w)
  with dz = (dU * g + q) * swish'(z) (dU: gradient of u, q[n,c]: gradient through the pooled mean).  Today one kernel makes
  a pass over (dU, y) for them (k_se_gate_bwd).  They factor into four [n][c] tables -- A1, A2 from the projection's
  backward epilogue (it holds dU), B1, B2 from the forward pooling pass -- so no such pass is needed:
      S1 = sum_n g * A1 + q * B1        S2 = sum_n g * A2 + q * B2."""
  rng = np.random.default_rng(0)
  n, hw, c = 3, 35, 8
  y = torch.tensor(rng.standard_normal((n, hw, c)) * 1.5 + 0.3)
  gamma = torch.tensor(1 + 0.3 * rng.standard_normal(c))
  beta = torch.tensor(0.2 * rng.standard_normal(c))
  w1 = torch.tensor(rng.standard_normal((c, 3)) * 0.5)
  w2 = torch.tensor(rng.standard_normal((3, c)) * 0.5)
  wp = torch.tensor(rng.standard_normal((c, 5)) * 0.4)
  d_out = torch.tensor(rng.standard_normal((n, hw, 5)))
  yq = y.clone().requires_grad_(True)
  mu = yq.mean((0, 1))
  rstd = torch.rsqrt(yq.var((0, 1), unbiased=False) + 1e-3)
  zhat = (yq - mu) * rstd
  z = zhat * gamma + beta
  z.retain_grad()
  h = z * torch.sigmoid(z)
  pooled = h.mean(1)
  s = pooled @ w1
  g = torch.sigmoid((s * torch.sigmoid(s)) @ w2)                     # [n, c]
  u = h * g[:, None, :]
  out = u @ wp
  out.backward(d_out)
  dz = z.grad                                                        # what k_se_gate_bwd writes
  s1_ref, s2_ref = dz.sum((0, 1)), (dz * zhat.detach()).sum((0, 1))
  # the pieces a kernel would have: dU = d_out Wp^T, q = gradient through the pooled mean (from the FC backward)
  with torch.no_grad():
    dU = d_out @ wp.T
    zz, zh = z.detach(), zhat.detach()
    sg = torch.sigmoid(zz)
    hh = zz * sg
    dphi = sg * (1 + zz * (1 - sg))
    dgate = (dU * hh).sum(1)                                         # [n, c]: the gate sums of the tiled backward (XM = 2)
  gq = g.detach().clone().requires_grad_(True)
  # pooled path: d pooled[n,c] = sum_c' dgate[n,c'] * d g[n,c'] / d pooled[n,c]
  pq = pooled.detach().clone().requires_grad_(True)
  s_ = pq @ w1
  g_ = torch.sigmoid((s_ * torch.sigmoid(s_)) @ w2)
  g_.backward(dgate)
  q = pq.grad / hw                                                   # added to every pixel of the image
  with torch.no_grad():
    a1 = (dU * dphi).sum(1)
    a2 = (dU * dphi * zh).sum(1)
    b1 = dphi.sum(1)
    b2 = (dphi * zh).sum(1)
    gd = g.detach()
    s1 = (gd * a1 + q * b1).sum(0)
    s2 = (gd * a2 + q * b2).sum(0)
    # and dz itself, formed on load by the consumer
    dz_on_load = (dU * gd[:, None, :] + q[:, None, :]) * dphi
  assert torch.allclose(dz_on_load, dz, rtol=1e-10, atol=1e-12)
  assert torch.allclose(s1, s1_ref, rtol=1e-10, atol=1e-12)
  assert torch.allclose(s2, s2_ref, rtol=1e-10, atol=1e-12)
