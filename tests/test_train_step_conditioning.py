"""Conditioning guard for tests/test_gpu_network.py::test_train_step_matches_oracle_fp32 (CPU only, oracle only).

The round-4 gate went red on a BISTABLE test case: `conv_after_downsample=True` at 128 px with two images pools a 4 x 4,
320-channel C5 whose top-2 window candidates are 3.5e-6 apart, so the max-pool argmax -- and with it up to 70 % of a
backbone tensor's gradient -- depends on the last bits of the forward pass.  The oracle's own gradients flipped by the same
191 of 493 tensors under a 1e-7 scaling of its input; no implementation can "agree with the oracle" on such a case.

This test makes that measurement the rule: for EVERY parametrised case of the device test the oracle's train step is run at
x, x (1 + 1e-7) and x (1 - 1e-7) -- less than one fp32 ulp for most pixels, both directions of every near-tie.  A case that
the device test checks tensor by tensor at 1e-2 must not move any tensor by more than 3e-3 (a third of that tolerance;
measured: <= 7e-4); the two documented carve-outs (d7x at a CPU-tractable size, an activation with a kink) must really be
ill conditioned, or they lose their carve-out.  A case that fails here gets another size / batch / seed, not an allowance.
"""
import numpy as np
import pytest
import torch

from oracle import efficientdet_oracle as orc
from tests import test_gpu_network as tgn

GUARD = 3e-3


class _HostDraws(dict):
  """Stochastic-depth draws with the device path's structure -- floor(p + u) / p per image, own draws per residual block
  / tower layer (utils.drop_connect, utils.py:329-344) -- from a seeded host generator: the device test hands the oracle
  the engine's draws; here every scope the oracle asks for gets one, the same in the perturbed runs (seeded per scope)."""

  def __init__(self, config, batch):
    super().__init__()
    self.batch = batch
    self.tower_p = getattr(config, 'survival_prob', None)
    spec = tgn.netspec.NetSpec(config)
    self.block_p = getattr(spec, 'survival_probs', None)

  def __contains__(self, key):
    if not dict.__contains__(self, key):
      if ':l' in key:
        p = self.tower_p
      else:
        p = self.block_p[int(key.rsplit('_', 1)[1])] if self.block_p else None
      if not p or p >= 1.0:
        return False
      import zlib
      u = np.random.default_rng(zlib.crc32(key.encode())).random(self.batch).astype(np.float32)
      self[key] = torch.from_numpy(np.floor(np.float32(p) + u) / np.float32(p)).float()
    return True


def _grads(case, eps):
  config, vals, images, labels = tgn.train_step_problem(case)
  oracle = orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
  x = torch.from_numpy(images)
  with torch.no_grad():
    oracle.forward(x[:1], False)                  # registers the trainable list
  oracle.drop_scale = _HostDraws(config, images.shape[0])
  tl = {k: torch.from_numpy(v) for k, v in labels.items()}
  _, grads = orc.train_step(oracle, x * (1.0 + eps), tl, {}, 0.02, 0.9)
  return {k: v.detach().double() for k, v in grads.items()}


def moved(g0, g1):
  """Per-tensor move on the device test's own yardstick: max |dg| / max(|g|_max, GRAD_FLOOR * largest |g|), the fusion
  scalars left out (the device test compares those as one vector)."""
  gmax = max(float(g.abs().max()) for g in g0.values())
  out = {}
  for k, g in g0.items():
    if k.rsplit('/', 1)[-1].startswith('WSM'):
      continue
    out[k] = float((g1[k] - g).abs().max()) / max(float(g.abs().max()), tgn.GRAD_FLOOR * gmax)
  return out


@pytest.mark.parametrize('case', tgn.TRAIN_STEP_CASES, ids=lambda c: '%s[%s]@%dx%d' % (c[0], c[1], c[2], c[3]))
def test_every_fp32_train_step_case_is_conditioned_or_carved_out(case):
  torch.set_num_threads(min(8, torch.get_num_threads()))
  g0 = _grads(case, 0.0)
  worst = {}
  for eps in (1e-7, -1e-7):
    for k, e in moved(g0, _grads(case, eps)).items():
      worst[k] = max(worst.get(k, 0.0), e)
  top = sorted(worst.items(), key=lambda t: -t[1])[:4]
  over = sum(e > GUARD for e in worst.values())
  print('%s: %d/%d tensors move by more than %g under a 1e-7 input scaling; worst %s' % (case, over, len(worst), GUARD, top))
  if tgn.train_step_case_is_ill_conditioned(case):
    assert over >= 10, 'this case no longer needs its carve-out in test_train_step_matches_oracle_fp32: %s' % (top,)
  else:
    assert over == 0, ('ill-conditioned test case (the oracle disagrees with itself): choose another size / batch / seed, '
                       '%d tensors beyond %g, worst %s' % (over, GUARD, top))


def test_the_round4_case_is_bistable():
  """The case that turned the round-4 gate red, kept as a known-answer test of the guard itself: two images at 128 px."""
  case = ('efficientdet-d0', 'conv_after_downsample=True', 128, 2)
  worst = moved(_grads(case, 0.0), _grads(case, 1e-7))
  assert sum(e > 1e-2 for e in worst.values()) >= 100 and max(worst.values()) >= 0.3, sorted(worst.values())[-4:]


def test_host_draws_reach_the_stochastic_depth_cases():
  """d1 (backbone stochastic depth) and the tower-residual case really run with dropped images in the guard."""
  for case, n in ((('efficientdet-d1', '', 192, 2), 16), (tgn.TOWER_SD_CASE, None)):
    config, vals, images, labels = tgn.train_step_problem(case)
    oracle = orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
    oracle.drop_scale = _HostDraws(config, images.shape[0])
    with torch.no_grad():
      oracle.forward(torch.from_numpy(images), True)
    assert len(oracle.drop_scale) == (n if n else 2 * 5 * (config.box_class_repeats - 1)), sorted(oracle.drop_scale)
    scales = torch.stack(list(oracle.drop_scale.values()))
    assert float(scales.max()) > 1.0, scales
