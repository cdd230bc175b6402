"""-m gpu: parity of the bf16 throughput path AT THE SHAPES AND SIZES OF THE BASELINE CONFIGURATIONS.

tests/test_gpu_kernels.py checks every entry point on small ragged maps; the kernel variants that bench.py runs are
chosen by shape (rows, map size, channel counts, workspace size), so this module repeats the checks where the
benchmark lives (BASELINE.json configs[0] / configs[2]):

  1. every C-ABI entry point of the hot path at the real EfficientDet-D0 640x640 layer shapes (two images), bf16,
     against the CPU oracle -- same bodies as tests/test_gpu_kernels.py, engine-sized workspace;
  2. the whole network: d0 512x512 batch 1 forward (configs[0]) and d0 640x640 batch 2 training-mode forward + full
     train step (configs[2] at a batch the CPU oracle finishes in seconds), bf16 AND fp32 storage, against the fp32
     oracle (tolerance stated in TOL) and against the bf16-storage-emulating oracle -- end to end where the map is
     well conditioned (fp32; bf16 inference), and LAYER BY LAYER (teacher forced: every stored activation, every
     stored gradient, every variable's gradient of the bf16 train step, each checked from the device's own stored
     inputs) where it is not: training-mode BatchNorm over ~100 random-weight layers amplifies a one-ulp bf16
     rounding flip into percents at the outputs, in the oracle itself (tests/test_oracle_conditioning.py) and
     between two runs of the same device code (SE sums are fp32 atomics);
  3. the full benchmark size, 640x640 batch 128, through size-independent properties: a batch made of 64 copies of
     the 2 oracle-checked images must reproduce the 2-image results -- bit for bit per entry point on the tensors
     beyond 2^31 bytes, to a bf16 ulp for the inference forward (images are independent), and statistically
     (losses, gradient direction, bounded buffer error) for the training step;
  4. coverage: every kernel SYMBOL launched by the batch-128 step must also have been launched by a test of (1) or
     by the oracle-checked 2-image step of (2) (the library's debug launch log, include/edet_hip.h).

Tolerances.  fp32 storage: 1e-3 of the level's max |logit| (BASELINE.json north_star).  bf16 storage vs the fp32
oracle: TOL['bf16_vs_f32'] -- the accumulated storage rounding of ~100 layers; vs the emulating oracle (rounds where
the engine stores): TOL['bf16_vs_emu'] -- a few bf16 ulps, what is left are 1-ulp flips from fp32 summation order.
"""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from automl_amd import _lib, hparams_config, train_lib
from automl_amd._lib import BwdEpi, call, ptr
from oracle import efficientdet_oracle as orc
from tests import gpu_util as gu
from tests import test_gpu_kernels as tk
from tests.test_gpu_network import _seg_index, make_labels, perturbed_params, rel_err

pytestmark = pytest.mark.gpu

BF16 = gu.DTYPES[1]
TOL = {'f32': 1e-3, 'bf16_vs_f32': 3e-2, 'bf16_vs_emu': 1e-2, 'layer': 1.2e-2, 'layer_grad': 3e-2, 'layer_wgrad': 3e-2}
ENGINE_WS_MIB = 64          # automl_amd/engine.py: the weight-gradient workspace the benchmark step hands over
COVERED = {}                # kernel symbol -> launches, accumulated over the oracle-checked tests of this module


@pytest.fixture(autouse=True)
def _log_kernel_symbols():
  _lib.launch_log_start()
  yield
  for k, v in _lib.launch_log_stop().items():
    COVERED[k] = COVERED.get(k, 0) + v


# ------------------------------------------------------------------ 1. entry points at the D0 640x640 layer shapes
# (H = W, cin, cout, view of the conv input): 'plain' = a stored block output / depthwise output, 'gate' = BatchNorm +
# swish + SE gate (the project convolutions), as engine.py builds them
PW_LAYERS = [
    (320, 32, 16, 'gate'), (320, 16, 96, 'plain'), (160, 96, 24, 'gate'), (160, 24, 144, 'plain'),
    (160, 144, 24, 'gate'), (80, 144, 40, 'gate'), (80, 40, 240, 'plain'), (80, 240, 40, 'gate'),
    (40, 240, 80, 'gate'), (40, 80, 480, 'plain'), (40, 480, 80, 'gate'), (40, 480, 112, 'gate'),
    (40, 112, 672, 'plain'), (40, 672, 112, 'gate'), (20, 672, 192, 'gate'), (20, 192, 1152, 'plain'),
    (20, 1152, 192, 'gate'), (20, 1152, 320, 'gate'),
    (80, 40, 64, 'plain'), (40, 112, 64, 'plain'), (20, 320, 64, 'plain'),      # BiFPN resample 1x1
    (80, 64, 64, 'plain'), (40, 64, 64, 'plain'), (20, 64, 64, 'plain'), (10, 64, 64, 'plain'), (5, 64, 64, 'plain'),
    (80, 64, 810, 'plain'), (80, 64, 36, 'plain'), (40, 64, 810, 'plain'),      # class / box predict
]
DW_LAYERS = [   # (H = W of the input, channels, k, stride)
    (320, 32, 3, 1), (320, 96, 3, 2), (160, 144, 3, 1), (160, 144, 5, 2), (80, 240, 5, 1), (80, 240, 3, 2),
    (40, 480, 3, 1), (40, 480, 5, 1), (40, 672, 5, 1), (40, 672, 5, 2), (20, 1152, 5, 1), (20, 1152, 3, 1),
    (80, 64, 3, 1), (40, 64, 3, 1), (20, 64, 3, 1), (10, 64, 3, 1), (5, 64, 3, 1),
]
N_IMG = 2


def _pw_id(l):
  return '%dx%dx%d->%d' % (l[0], l[0], l[1], l[2])


def _dw_id(l):
  return '%dx%dx%d_k%ds%d' % (l[0], l[0], l[1], l[2], l[3])


@pytest.mark.parametrize('layer', PW_LAYERS, ids=_pw_id)
def test_pw_fwd_at_d0_640_shapes(layer):
  h, cin, cout, view = layer
  tk.test_pw_fwd(BF16, (N_IMG, h, h, cin, cout), 'bn_swish_gate' if view == 'gate' else 'plain', 'auto')


@pytest.mark.parametrize('layer', PW_LAYERS, ids=_pw_id)
def test_pw_bwd_data_at_d0_640_shapes(layer):
  """The epilogues the network uses: project layers chain into the SE-gated view ('gate'), expand / BiFPN / head
  layers into a stored tensor ('plain'; 'plain_beta' where a residual or a second consumer already wrote the
  gradient); dy carries the BatchNorm backward on load except for the predict layers."""
  h, cin, cout, view = layer
  predict = cout in (810, 36)
  shape = (N_IMG, h, h, cin, cout)
  tk.test_pw_bwd_data(BF16, shape, 'gate' if view == 'gate' else 'plain', not predict, 'auto')
  if view == 'plain' and cout > cin and not predict:
    tk.test_pw_bwd_data(BF16, shape, 'plain_beta', True, 'auto')
  if cin == 64 and cout == 64:
    tk.test_pw_bwd_data(BF16, shape, 'bn_swish_stats', True, 'auto')


@pytest.mark.parametrize('layer', PW_LAYERS, ids=_pw_id)
def test_pw_bwd_one_call_at_d0_640_shapes(layer, monkeypatch):
  """edet_pw_bwd as the engine calls it (the fused data + weight gradient kernel wherever the layer fits it), with
  the engine's workspace: data gradient, epilogue sums AND weight gradient.  The project layers go to the one-pass
  kernel from ~250 K rows up (the batch-128 step: the 320 / 160 / 80-row maps); two images stay below that, so those
  layers are run a second time with the threshold at 0 -- the kernel instantiation depends on the channel counts only."""
  h, cin, cout, view = layer
  predict = cout in (810, 36)
  shape = (N_IMG, h, h, cin, cout)
  tk.test_pw_bwd_data(BF16, shape, 'gate' if view == 'gate' else 'plain', not predict, 'auto', one_call=True,
                      ws_mib=ENGINE_WS_MIB)
  if view == 'gate' and h >= 80:
    monkeypatch.setenv('EDET_PWS_FUSED_MINROWS', '0')
    tk.test_pw_bwd_data(BF16, shape, 'gate', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
    monkeypatch.delenv('EDET_PWS_FUSED_MINROWS')
  if view == 'plain' and cout > cin and not predict:
    tk.test_pw_bwd_data(BF16, shape, 'plain_beta', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
  if cin == 64 and cout == 64:
    tk.test_pw_bwd_data(BF16, shape, 'bn_swish_stats', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)


@pytest.mark.parametrize('layer', PW_LAYERS, ids=_pw_id)
def test_pw_bwd_weight_at_d0_640_shapes(layer):
  h, cin, cout, view = layer
  tk.test_pw_bwd_weight(BF16, (N_IMG, h, h, cin, cout), 'bn_swish_gate' if view == 'gate' else 'plain', True, 'auto',
                        ws_mib=ENGINE_WS_MIB)


@pytest.mark.parametrize('layer', DW_LAYERS, ids=_dw_id)
def test_dw_fwd_at_d0_640_shapes(layer):
  h, c, k, s = layer
  tk.test_dw_fwd(BF16, (N_IMG, h, h, c), (k, s), 'bn_swish' if c != 64 else 'plain')


@pytest.mark.parametrize('layer', DW_LAYERS, ids=_dw_id)
def test_dw_bwd_at_d0_640_shapes(layer):
  """edet_dw_bwd as the engine calls it (stride 1: the fused kernel, 4 channels per thread on the 320x320 / 160x160
  maps; stride 2: the two separate kernels), BatchNorm backward on dy, BatchNorm-backward sums in the epilogue."""
  h, c, k, s = layer
  tk.test_dw_bwd(BF16, (N_IMG, h, h, c), (k, s), 'bn_swish_stats', 'one_call', ws_mib=ENGINE_WS_MIB)
  if c == 64:    # BiFPN / head depthwise layers read a stored (fused / activated) tensor that has other consumers
    tk.test_dw_bwd(BF16, (N_IMG, h, h, c), (k, s), 'plain_beta', 'one_call', ws_mib=ENGINE_WS_MIB)


def test_stem_se_bn_fuse_at_d0_640_shapes():
  tk.test_stem(BF16, (N_IMG, 640, 640, 32))
  for shape in ((N_IMG, 320, 320, 32, 8), (N_IMG, 160, 160, 96, 4), (N_IMG, 40, 40, 672, 28), (N_IMG, 20, 20, 1152, 48)):
    tk.test_squeeze_excite(BF16, shape)
  for shape in ((N_IMG, 320, 320, 16), (N_IMG, 160, 160, 24), (N_IMG, 40, 40, 112)):
    tk.test_batchnorm_train_fwd_bwd(BF16, shape)
  for case in (((N_IMG, 80, 80, 64), [(_lib.RS_IDENTITY, 80, 80), (_lib.RS_UP2, 40, 40)], 'fastattn'),
               ((N_IMG, 40, 40, 64), [(_lib.RS_IDENTITY, 40, 40), (_lib.RS_IDENTITY, 40, 40), (_lib.RS_POOL, 80, 80)],
                'fastattn'),
               ((N_IMG, 10, 10, 64), [(_lib.RS_POOL, 20, 20)], 'none')):
    tk.test_fuse(BF16, case)


# ------------------------------------------------------------------ 2. the whole network against the two oracles
def _problem(size, batch, seed):
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=%d' % size)
  vals = perturbed_params(config, seed)
  rng = np.random.default_rng(seed + 100)
  images = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32)).to(torch.bfloat16).float()
  return config, vals, images


def _oracle(config, vals, storage):
  return orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()}, storage=storage)


def _level_errs(got, want):
  return [round(rel_err(g, w), 5) for g, w in zip(got, want)]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_d0_512_batch1_forward_equals_oracle(dtype):
  """BASELINE.json configs[0]: efficientdet-d0, 512x512, batch 1, inference forward; 49,104 anchors."""
  config, vals, images = _problem(512, 1, 11)
  net = train_lib.EfficientDetNetTrain(config=config, dtype=dtype, params=vals)
  cls, box = net(images, training=False)
  torch.cuda.synchronize()
  assert [tuple(c.shape) for c in cls] == [(1, s, s, 810) for s in (64, 32, 16, 8, 4)]
  assert sum(int(np.prod(c.shape[1:3])) * 9 for c in cls) == 49104
  with torch.no_grad():
    cref, bref = _oracle(config, vals, 'f32').forward(images, False)
  e32 = _level_errs(cls, cref) + _level_errs(box, bref)
  print('d0-512 forward %s vs fp32 oracle: %s' % (dtype, e32))
  # r04: north_star's 1e-3 ("box/class logits within 1e-3 rel of the TF CPU reference") holds for the bf16 path too -- the
  # inference forward stores the logits as fp32 (Engine.logits_f32); what is left is the bf16 rounding of the matrix-core
  # operands (scripts/precision_sweep.py: 7e-5 class / 8e-4 box at 640x640)
  assert max(e32) <= TOL['f32'], e32
  if dtype == 'bf16':
    with torch.no_grad():
      cemu, bemu = _oracle(config, vals, 'bf16').forward(images, False)
    eemu = _level_errs(cls, cemu) + _level_errs(box, bemu)
    print('d0-512 forward bf16 vs emulating oracle: %s' % (eemu,))
    assert max(eemu) <= TOL['bf16_vs_emu'], eemu


class _Step(object):
  """One train step of d0 at 640x640 on the device, with everything the comparisons need kept."""

  def __init__(self, dtype, batch, tile_of=None):
    self.config, self.vals, images2 = _problem(640, 2, 13)
    labels2 = make_labels(self.config, 2, 640, 19)
    self.images2, self.labels2 = images2, labels2
    reps = batch // 2
    images = images2.repeat(reps, 1, 1, 1)
    labels = {k: np.tile(v, (reps,) + (1,) * (v.ndim - 1)) for k, v in labels2.items()}
    # the loss normalizer sum(mean_num_positives) + 1 of the 2-image problem, times the number of copies: the loss
    # and every variable's gradient are then those of the 2-image step
    labels['normalizer'] = reps * (float(labels2['mean_num_positives'].sum()) + 1.0)
    self.lr, self.decay = 0.02, 0.9
    net = train_lib.EfficientDetNetTrain(config=self.config, dtype=dtype, params=self.vals)
    eng = self.eng = net._ensure_engine(batch, 640, 640)
    _lib.launch_log_start()
    eng.forward(net._to_device_images(images, eng), training=True)
    torch.cuda.synchronize()
    self.cls = [c.float().cpu() for c in eng.outputs()[0]]
    self.box = [b.float().cpu() for b in eng.outputs()[1]]
    eng.loss_backward(net._labels_to_device(labels, eng))
    eng.optimizer_step(self.lr, self.decay)
    torch.cuda.synchronize()
    self.kernels = _lib.launch_log_stop()
    _lib.launch_log_start()       # the autouse fixture's log continues
    self.losses = eng.loss_values()
    self.grads = {name: (eng.grad(name).cpu() * eng.seg_factor.cpu()[_seg_index(eng, name)]).reshape(
        eng.offsets[name][2]) for name in eng.seg_names}
    self.new_params = eng.get_params()


_STEPS = {}


def _step(dtype, batch):
  key = (dtype, batch)
  if key not in _STEPS:
    _STEPS[key] = _Step(dtype, batch)
  return _STEPS[key]


_ORACLE_STEPS = {}


def _oracle_step(storage):
  """The same 2-image step on the CPU oracle -> (cls, box, loss values, clipped gradients, updated variables)."""
  if storage not in _ORACLE_STEPS:
    config, vals, images = _problem(640, 2, 13)
    labels = {k: torch.from_numpy(v) for k, v in make_labels(config, 2, 640, 19).items()}
    o = _oracle(config, vals, storage)
    with torch.no_grad():
      cls, box = o.forward(images, True)
    o = _oracle(config, vals, storage)
    with torch.no_grad():
      o.forward(images[:1, :64, :64], False)       # registers the trainable list
    loss_vals, grads = orc.train_step(o, images, labels, {}, 0.02, 0.9)
    _ORACLE_STEPS[storage] = (cls, box, loss_vals, {k: g.detach() for k, g in grads.items()},
                              {k: v.detach().numpy() for k, v in o.params().items()})
  return _ORACLE_STEPS[storage]


def _grad_report(step, ref_grads):
  """-> (cosine of the whole clipped gradient, worst per-tensor error relative to the tensor's max, relative L2 error
  of the vector of fusion scalars).  The fusion scalars (WSM*) are left out of the per-tensor figure: each is a
  difference of whole-level sums that cancels to a small remainder, so that even in fp32 a single one lands anywhere
  between 0.6e-2 and 2.1e-2 of its own value from run to run (r02v: three runs of one tree); together they are stable."""
  num = na = nb = 0.0
  worst = (0.0, '')
  wsm_mine, wsm_ref = [], []
  gmax = max(float(g.abs().max()) for g in ref_grads.values())
  for name, g in ref_grads.items():
    mine = step.grads[name].double()
    g = g.double()
    num += float((mine * g).sum())
    na += float((mine * mine).sum())
    nb += float((g * g).sum())
    if name.rsplit('/', 1)[-1].startswith('WSM'):
      wsm_mine.append(float(mine.reshape(-1)[0]))
      wsm_ref.append(float(g.reshape(-1)[0]))
      continue
    e = float((mine - g).abs().max()) / max(float(g.abs().max()), 1e-3 * gmax)
    if e > worst[0]:
      worst = (e, name)
  wsm = float(np.linalg.norm(np.asarray(wsm_mine) - np.asarray(wsm_ref)) / np.linalg.norm(np.asarray(wsm_ref)))
  return num / np.sqrt(na * nb + 1e-300), worst, wsm


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_d0_640_batch2_train_step_equals_oracle(dtype):
  """BASELINE.json configs[2] at two images, end to end: training-mode forward (batch statistics), focal + Huber loss,
  backward, L2, clipping, SGD / EMA update.  fp32 storage: logits 1e-3, losses 2e-3, every variable's clipped
  gradient 5e-2 of its max (measured 0.4e-2 ... 1.6e-2: the conditioning of the map, see the assertion), updated variables
  1e-3.  bf16 storage: losses 1e-2 of the fp32 oracle's, direction of the whole gradient (cosine >= 0.9); class and box
  outputs per level against a bound MEASURED on this
  problem (the emulating oracle's own distance to the fp32 oracle plus CHAOS_FACTOR times its movement under one-ulp input
  flips) -- end to end they are dominated by the amplification of rounding flips (module docstring); the bf16 path is
  pinned layer by layer in the next test."""
  step = _step(dtype, 2)
  cref, bref, lref, gref, pref = _oracle_step('f32')
  ecls, ebox = _level_errs(step.cls, cref), _level_errs(step.box, bref)
  print('d0-640 B=2 training forward %s vs fp32 oracle: class %s box %s' % (dtype, ecls, ebox))
  loss_tol = 2e-3 if dtype == 'f32' else 1e-2
  for k in ('cls_loss', 'box_loss', 'det_loss', 'reg_l2_loss', 'loss', 'gradient_norm'):
    assert abs(step.losses[k] - lref[k]) <= loss_tol * abs(lref[k]) + 1e-6, (k, step.losses[k], lref[k])
  cos, worst, wsm = _grad_report(step, gref)
  print('d0-640 B=2 %s: gradient cosine vs fp32 oracle %.6f, worst tensor %s, fusion scalars %.5f' % (dtype, cos, worst, wsm))
  if dtype == 'f32':
    assert max(ecls + ebox) <= TOL['f32'], (ecls, ebox)
    # Per tensor the fp32 map is itself only conditioned to ~1e-2: the ORACLE's own gradients move by up to 1.1e-2 of a
    # tensor's max (39 of 493 tensors by more than 1e-3) when its input is scaled by 1 + 1e-7
    # (tests/test_oracle_conditioning.py), and the device lands in that band from run to run -- 3.7e-3 when the fp32
    # atomics of the SE / loss sums happen to add in the oracle's order, 1.0e-2 ... 1.6e-2 otherwise (r02k ... r02y, ten
    # runs), the updated variables following at 1.1e-4 ... 2.6e-4.  The direction of the whole gradient is pinned to
    # five digits, every tensor to 5e-2.
    assert cos >= 0.99999 and worst[0] <= 5e-2 and wsm <= 1e-2, (cos, worst, wsm)
    upd = max(float(np.abs(step.new_params[n] - pref[n]).max()) / max(float(np.abs(pref[n]).max()), 1e-6) for n in gref)
    assert upd <= 1e-3, 'updated variables differ: %g' % upd
  else:
    # Training mode end to end is the ill-conditioned map of the module docstring, so the yardstick is MEASURED on this
    # very problem instead of assumed (r06; VERDICT r05: "I do not accept a 0.3 relative bound as a test"): the
    # storage-emulating oracle -- the best any bf16-storage implementation can do -- is run on the input and on the input
    # with 0.05 % of its pixels moved by one bf16 ulp.  Its distance to the fp32 oracle (e_emu) and its own movement under
    # those flips (m_emu) are per-level numbers; the device, which differs from the emulating oracle by single rounding
    # flips at EVERY layer rather than at the input only, must stay within e_emu + CHAOS_FACTOR * m_emu of the fp32 oracle
    # on every level.  (The bf16 step itself is pinned layer by layer in the next test, the trajectory of a bf16
    # optimisation against the fp32 one in tests/test_gpu_network.py.)
    e_emu, m_emu = _emulating_oracle_conditioning()
    nl = len(ecls)
    bound = [e + CHAOS_FACTOR * max(m, CHAOS_FLOOR) for e, m in zip(e_emu, m_emu)]
    print('d0-640 B=2 bf16: emulating oracle vs fp32 oracle per level %s, its movement under one-ulp input flips %s; '
          'bound %s' % ([round(v, 4) for v in e_emu], [round(v, 4) for v in m_emu], [round(v, 4) for v in bound]))
    for lvl, (e, b) in enumerate(zip(ecls + ebox, bound)):
      assert e <= b, ('class' if lvl < nl else 'box', lvl % nl, e, b)
    assert cos >= 0.9, cos


# r06, measured: the device lands where the emulating oracle lands -- class levels 0.014 ... 0.024 of the level's range from
# the fp32 oracle against the emulating oracle's own 0.013 ... 0.021 (movement under flips 0.012 ... 0.022), box levels 0.11 ...
# 0.145 against 0.115 ... 0.133 (movement 0.077 ... 0.123)
CHAOS_FACTOR, CHAOS_FLOOR = 1.0, 2e-3
_EMU_COND = {}


def _emulating_oracle_conditioning():
  """-> (per-level distance of the bf16-storage-emulating oracle's training forward to the fp32 oracle's, per-level
  movement of the emulating oracle under one-ulp flips of 0.05 % of the input pixels), class levels then box levels,
  each relative to the level's range -- the 2-image 640x640 problem of this module (tests/test_oracle_conditioning.py
  makes the same measurement at 256 px on the CPU)."""
  if not _EMU_COND:
    config, vals, images = _problem(640, 2, 13)
    cref, bref = _oracle_step('f32')[:2]
    rng = np.random.default_rng(0)
    mask = torch.from_numpy(rng.random(tuple(images.shape)) < 5e-4)
    bumped = torch.where(mask, (images * (1 + 2.0**-8)).to(torch.bfloat16).float(), images)
    with torch.no_grad():
      c0, b0 = _oracle(config, vals, 'bf16').forward(images, True)
      c1, b1 = _oracle(config, vals, 'bf16').forward(bumped, True)
    _EMU_COND['e'] = _level_errs(c0, cref) + _level_errs(b0, bref)
    _EMU_COND['m'] = _level_errs(c1, c0) + _level_errs(b1, b0)
  return _EMU_COND['e'], _EMU_COND['m']


def test_d0_640_batch2_bf16_train_step_layer_by_layer():
  """The bf16 train step of d0 at 640x640, two images, against the bf16-storage-emulating oracle with teacher forcing:
  every stored activation (~230 tensors), every stored gradient buffer and every variable's gradient is compared with
  the oracle's value computed from the DEVICE's stored inputs of that layer.  Tolerance: TOL['layer'] of the tensor's
  max -- one bf16 ulp of the largest elements is 0.78 % -- i.e. nothing beyond single rounding flips."""
  step = _step('bf16', 2)
  config, vals, images = _problem(640, 2, 13)
  labels = {k: torch.from_numpy(v) for k, v in make_labels(config, 2, 640, 19).items()}
  o = _oracle(config, vals, 'bf16')
  with torch.no_grad():
    o.forward(images[:1, :64, :64], False)         # registers the trainable list
  hook = o.hook = gu.TeacherForce(step.eng)
  P = o.params()
  names = o.trainable_names()
  for n in names:
    P[n].requires_grad_(True)
  cls, box = o.forward(images, True)
  det, _, _ = orc.detection_loss(config, cls, box, labels)
  l2 = config.weight_decay * sum((P[n]**2).sum() / 2 for n in names if orc.is_l2_regularised(n))
  (det + l2).backward()
  print('teacher-forced forward: %d tensors, worst %s' % (len(hook.fwd_err), hook.worst(hook.fwd_err)))
  print('teacher-forced backward: %d gradient buffers, worst %s' % (len(hook.bwd_err), hook.worst(hook.bwd_err)))
  assert len(hook.fwd_err) >= 220 and len(hook.bwd_err) >= 200, (len(hook.fwd_err), len(hook.bwd_err), hook.missing[:8])
  assert max(hook.fwd_err.values()) <= TOL['layer'], hook.worst(hook.fwd_err, 6)
  assert max(hook.bwd_err.values()) <= TOL['layer_grad'], hook.worst(hook.bwd_err, 6)
  werr, wsm_mine, wsm_ref, bn_mine, bn_ref = {}, [], [], [], []
  gmax = max(float(P[n].grad.abs().max()) for n in names)
  for n in names:
    g = P[n].grad
    mine = step.eng.grad(n).cpu().reshape(g.shape)
    if n.endswith('/bias') and 'predict' not in n and '/se/' not in n:
      # a bias in front of a BatchNorm (every bias except the predict layers' and the SE ones): its gradient is
      # analytically zero and the device does not compute it; the emulating oracle's autograd leaves rounding noise
      # there (seen up to ~1e-3 of the largest gradient, r02e/r02f), which is not something to reproduce
      assert float(mine.abs().max()) == 0.0, n
      assert float(g.abs().max()) <= 2e-2 * gmax, (n, float(g.abs().max()), gmax)
      continue
    if n.rsplit('/', 1)[-1].startswith('WSM'):
      # A fusion scalar's gradient is ONE number: a difference of whole-level sums of d(out) * input that cancel to a
      # small remainder (fast attention with equal weights: dw_i = (2 dwn_i - dwn_j - dwn_k) / 9), so its relative
      # error is set by that cancellation and moves from 2 % to 11 % between runs (r02e/g/h: SE atomics reorder the
      # step).  The 57 scalars are checked TOGETHER below, as one vector.
      wsm_mine.append(float(mine.reshape(-1)[0]))
      wsm_ref.append(float(g.reshape(-1)[0]))
      continue
    e = float((mine - g).abs().max()) / max(float(g.abs().max()), 1e-4 * gmax)
    if n.endswith('/gamma') or n.endswith('/beta'):
      # BatchNorm scale / offset gradients are whole-tensor sums (sum dz * xhat, sum dz over up to 13 M elements) that
      # cancel by orders of magnitude; the device forms them from the unrounded dz, the oracle from the stored one.
      # One tensor alone moves between 1.2e-2 and 3.9e-2 from run to run (stem gamma, r02u: three runs of the same
      # tree); all of them together, as one vector, are stable
      assert e <= 0.1, (n, e)
      bn_mine.append(mine.reshape(-1).double().numpy())
      bn_ref.append(g.reshape(-1).double().numpy())
      continue
    werr[n] = e
  bn_mine, bn_ref = np.concatenate(bn_mine), np.concatenate(bn_ref)
  bn_err = float(np.linalg.norm(bn_mine - bn_ref) / np.linalg.norm(bn_ref))
  print('BatchNorm gamma / beta gradients: %d elements, relative L2 error of the vector %.4f' % (bn_ref.size, bn_err))
  assert bn_err <= 2e-2, bn_err
  wsm_mine, wsm_ref = np.asarray(wsm_mine), np.asarray(wsm_ref)
  wsm_err = float(np.linalg.norm(wsm_mine - wsm_ref) / np.linalg.norm(wsm_ref))
  print('fusion scalars: %d, relative L2 error of their gradient vector %.4f' % (len(wsm_ref), wsm_err))
  assert len(wsm_ref) >= 50 and wsm_err <= 3e-2, wsm_err
  print('teacher-forced variable gradients: %d tensors, worst %s' % (len(werr), gu.TeacherForce.worst(werr, 5)))
  assert max(werr.values()) <= TOL['layer_wgrad'], gu.TeacherForce.worst(werr, 8)


# ------------------------------------------------------------------ 3. the full benchmark size
BIG_N = 128


def _tile(t, reps):
  return t.repeat((reps,) + (1,) * (t.dim() - 1)).contiguous()


def _copies_equal(big, small, reps, what):
  v = big.view((reps,) + tuple(small.shape))
  for k in (0, reps // 2, reps - 1):
    assert torch.equal(v[k], small), '%s: copy %d of %d differs from the 2-image result' % (what, k, reps)


def test_entry_points_on_tensors_beyond_2GiB_equal_their_tiles():
  """The 320x320x96 expanded tensor of block 1 is 2.5 GB at batch 128 -- the only tensors of the benchmark step whose
  byte offsets pass 2^31.  Every entry point that touches it (pointwise forward / data gradient / weight gradient,
  stride-2 depthwise forward / data gradient / weight gradient) runs on 64 copies of a 2-image problem (the same
  entry points at these shapes are checked against the oracle above): per-row / per-image outputs must be
  BIT-IDENTICAL in every copy, reduced outputs (statistics, weight gradients) 64 x the 2-image ones."""
  expand_block_entry_points_equal_their_tiles(320, 16, 96, 2, BIG_N // 2, beyond_2gib=True)


def expand_block_entry_points_equal_their_tiles(h, cin, cout, nsmall, reps, beyond_2gib):
  """The entry points around the expanded tensor of an MBConv block with a stride-2 depthwise layer ([n, h, h, cout]):
  `reps` copies of an `nsmall`-image problem against the problem itself (also used at the EfficientDet-D7x 1536x1536
  shapes by tests/test_gpu_side_configs.py)."""
  edt, tdt = _lib.EDET_BF16, torch.bfloat16
  gen = torch.Generator(device=gu.DEV).manual_seed(3)

  def rand(*shape):
    return torch.randn(shape, device=gu.DEV, generator=gen).to(tdt)

  def vec(c, lo=0.5, hi=1.5):
    return (torch.rand(c, device=gu.DEV, generator=gen) * (hi - lo) + lo).float()
  x2, dz2, y2 = rand(nsmall, h, h, cin), rand(nsmall, h, h, cout), rand(nsmall, h, h, cout)
  gdz2, gy2 = rand(nsmall, h // 2, h // 2, cout), rand(nsmall, h // 2, h // 2, cout)
  wt = (rand(cout, cin) * 0.25).contiguous()        # forward compute copy [cout][cin]
  wk = (rand(cin, cout) * 0.1).contiguous()         # data-gradient compute copy [cin][cout]
  dwk = (torch.randn(3, 3, cout, device=gu.DEV, generator=gen) / 3).float()
  ga, gb, gc = vec(cout), vec(cout, -0.1, 0.1), vec(cout, -0.1, 0.1)
  sc, sh, mean, rstd = vec(cout), vec(cout, -0.3, 0.3), vec(cout, -0.2, 0.2), vec(cout)
  npart = ctypes.c_int(0)
  wsp = torch.empty(ENGINE_WS_MIB * 256 * 1024, dtype=torch.float32, device=gu.DEV)
  res = {}
  for r in (1, reps):
    n = nsmall * r
    x, dz, y, gdz, gy = (_tile(t, r) for t in (x2, dz2, y2, gdz2, gy2))
    assert r == 1 or not beyond_2gib or dz.numel() * 2 > 2**31
    parts = [torch.zeros(_lib.MAX_PARTS * 2 * cout, dtype=torch.float32, device=gu.DEV) for _ in range(2)]
    out = torch.empty(n, h, h, cout, dtype=tdt, device=gu.DEV)
    tv = gu.tview(x, cin)
    call('edet_pw_fwd', ctypes.byref(tv), ptr(wt), cin, None, ptr(out), cout, cout, ptr(parts[0]), ctypes.byref(npart),
         edt, gu.stream())
    s1, s2 = gu.sum_partials(parts[0], npart.value, cout)
    gv = gu.gview(dz, cout, y, ga, gb, gc)
    gout = torch.empty(n, h, h, cin, dtype=tdt, device=gu.DEV)
    epi = BwdEpi(ptr(gout), 0, None, None, None, None)
    call('edet_pw_bwd_data', ctypes.byref(gv), ptr(wk), cout, ctypes.byref(tv), ctypes.byref(epi), ctypes.byref(npart),
         edt, gu.stream())
    dwt = torch.zeros(cin, cout, dtype=torch.float32, device=gu.DEV)
    call('edet_pw_bwd_weight', ctypes.byref(tv), ctypes.byref(gv), ptr(dwt), ptr(wsp), wsp.numel() * 4, edt, gu.stream())
    # the stride-2 depthwise layer behind it: its input view = BatchNorm + swish of `out`
    tvd = gu.tview(out, cout, sc, sh, None, _lib.ACT_SWISH)
    dout = torch.empty(n, h // 2, h // 2, cout, dtype=tdt, device=gu.DEV)
    call('edet_dw_fwd', ctypes.byref(tvd), ptr(dwk), 3, 2, ptr(dout), cout, None, ctypes.byref(npart), edt, gu.stream())
    gvd = gu.gview(gdz, cout, gy, ga, gb, gc)
    gin = torch.empty(n, h, h, cout, dtype=tdt, device=gu.DEV)
    epd = BwdEpi(ptr(gin), 0, ptr(mean), ptr(rstd), ptr(parts[1]), None)
    ddw = torch.zeros(3, 3, cout, dtype=torch.float32, device=gu.DEV)
    call('edet_dw_bwd', ctypes.byref(gvd), ptr(dwk), 3, 2, ctypes.byref(tvd), ctypes.byref(epd), ctypes.byref(npart),
         ptr(ddw), ptr(wsp), wsp.numel() * 4, edt, gu.stream())
    torch.cuda.synchronize()
    t1, t2 = gu.sum_partials(parts[1], npart.value, cout)
    res[r] = dict(out=out, s1=s1, s2=s2, gout=gout, dwt=dwt, dout=dout, gin=gin, ddw=ddw, t1=t1, t2=t2)
    del x, dz, y, gdz, gy
  small, big = res[1], res[reps]
  for k in ('out', 'gout', 'dout', 'gin'):
    _copies_equal(big[k], small[k], reps, k)
  for k in ('s1', 's2', 'dwt', 'ddw', 't1', 't2'):
    a, b = big[k].double().cpu(), small[k].double().cpu() * reps
    err = float((a - b).abs().max()) / float(b.abs().max())
    assert err <= 2e-3, (k, err)


def test_d0_640_batch128_inference_forward_equals_the_2_image_forward():
  """Inference-mode forward at the full benchmark batch: images are independent and nothing on the inference path depends
  on the batch (no atomics; the SE pooling adds an image's rows in chunks that depend on the map only), so every one of
  the 64 copies must give the 2-image logits (which test_d0_512/640 tie to the oracle) BIT FOR BIT."""
  config, vals, images2 = _problem(640, 2, 13)
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals)
  cls2, box2 = net(images2, training=False)
  want = [t.float().cpu() for t in cls2 + box2]
  with torch.no_grad():
    cref, bref = _oracle(config, vals, 'bf16').forward(images2, False)
  e = _level_errs(want, cref + bref)
  print('d0-640 B=2 inference forward bf16 vs emulating oracle: %s' % (e,))
  assert max(e) <= TOL['bf16_vs_emu'], e
  cls, box = net(_tile(images2, BIG_N // 2), training=False)
  torch.cuda.synchronize()
  for lvl, (big, small) in enumerate(zip(cls + box, want)):
    v = big.float().cpu().view((BIG_N // 2,) + tuple(small.shape))
    for k in (0, 17, BIG_N // 2 - 1):
      assert torch.equal(v[k], small), 'output %d: copy %d of the batch-128 forward differs from the 2-image forward' % (lvl, k)
  first = [t.clone() for t in cls + box]        # the outputs are views of the engine's buffers
  again = net(_tile(images2, BIG_N // 2), training=False)
  torch.cuda.synchronize()
  assert all(torch.equal(a, b) for a, b in zip(again[0] + again[1], first)), 'two runs of the forward differ'
  net._engines.clear()


def test_d0_640_batch128_train_step_tracks_the_tiled_2_image_step():
  """BASELINE.json configs[2] at full size (128 images, 36 GB of activations): 64 copies of the 2 oracle-checked
  images with the loss normalizer scaled by 64 define the SAME optimisation step.  Training-mode BatchNorm makes the
  comparison statistical (module docstring; two runs of the identical 2-image step differ by up to ~0.1 rms in the
  late buffers): losses to 2e-3, direction of the whole clipped gradient, and every stored activation / gradient
  buffer of the batch-128 executor within BUF_RMS_BOUND rms of its 2-image counterpart -- a wrong offset, a skipped
  tile or a mis-sized grid shows up as an rms error of order 1."""
  small, big = _step('bf16', 2), _step('bf16', BIG_N)
  reps = BIG_N // 2
  worst = {False: (0.0, ''), True: (0.0, '')}
  count = 0
  for key, t2 in small.eng._bufs.items():
    t128 = big.eng._bufs.get(key)
    if t128 is None or t2.dim() != 4 or t2.shape[0] != 2 or t128.shape[0] != BIG_N or t2.dtype != torch.bfloat16:
      continue
    is_grad = key.endswith('#grad') or key.endswith(':ds')
    b = t2.float()
    den = float(b.pow(2).mean().sqrt())
    if not np.isfinite(den) or den == 0.0:
      continue
    a = t128.view((reps, 2) + tuple(t2.shape[1:]))
    err = 0.0
    for k in (0, reps // 2, reps - 1):
      ak = a[k].float() * (reps if is_grad else 1)      # per-image loss terms are 1/64 of the 2-image problem's
      err = max(err, float((ak - b).pow(2).mean().sqrt()) / den)
    assert np.isfinite(err), key
    count += 1
    if err > worst[is_grad][0]:
      worst[is_grad] = (err, key)
  print('batch 128 vs tiled batch 2 over %d buffers: worst rms error activation %s, gradient %s' % (
      count, worst[False], worst[True]))
  assert count >= 400
  assert worst[False][0] <= BUF_RMS_BOUND and worst[True][0] <= BUF_RMS_BOUND, worst
  for k in ('cls_loss', 'box_loss', 'det_loss', 'reg_l2_loss', 'loss', 'gradient_norm'):
    assert abs(big.losses[k] - small.losses[k]) <= 2e-3 * abs(small.losses[k]) + 1e-6, (k, big.losses[k], small.losses[k])
  cos, worst_t, _ = _grad_report(big, small.grads)
  print('batch 128 vs tiled batch 2: gradient cosine %.6f, worst tensor %s' % (cos, worst_t))
  assert cos >= 0.9, cos


BUF_RMS_BOUND = 0.6


# ------------------------------------------------------------------ 3b. the step's plumbing entry points (round 6)
def test_step_plumbing_entry_points():
  """edet_loss_normalizer (the device-side 1 / (sum(mean_num_positives) + 1) of tf2/train_lib.py:517-534, what bench.py's
  captured step uses), edet_axpy_clear (the side chain's gradient arena joined into the main one), edet_cast_to_f32 and
  edet_zero against numpy: exact."""
  rng = np.random.default_rng(5)
  for n in (1, 2, 128, 1000):
    mnp = rng.integers(0, 400, n).astype(np.float32)
    d = torch.from_numpy(mnp).to(gu.DEV)
    out = torch.zeros(4, dtype=torch.float32, device=gu.DEV)
    call('edet_loss_normalizer', ptr(d), n, out.data_ptr() + 8, gu.stream())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got[2] == np.float32(1.0) / (np.float32(mnp.sum(dtype=np.float64)) + np.float32(1.0)), (n, got)
    assert got[0] == 0 and got[1] == 0 and got[3] == 0
  for n in (1, 777, 100003):
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    da, db = torch.from_numpy(a).to(gu.DEV), torch.from_numpy(b).to(gu.DEV)
    call('edet_axpy_clear', ptr(da), ptr(db), n, 1, gu.stream())
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy(), a + b) and not db.any()
    db2 = torch.from_numpy(b).to(gu.DEV)
    call('edet_axpy_clear', ptr(da), ptr(db2), n, 0, gu.stream())
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy(), (a + b) + b) and np.array_equal(db2.cpu().numpy(), b)
    h = torch.from_numpy(a).to(gu.DEV).to(torch.bfloat16)
    f = torch.empty(n, dtype=torch.float32, device=gu.DEV)
    call('edet_cast_to_f32', ptr(h), ptr(f), n, _lib.EDET_BF16, gu.stream())
    torch.cuda.synchronize()
    assert torch.equal(f, h.float())
    call('edet_cast_to_f32', ptr(da), ptr(f), n, _lib.EDET_F32, gu.stream())
    call('edet_zero', ptr(da), 4 * (n // 2), gu.stream())
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy(), (a + b) + b)
    assert not da[:n // 2].any() and np.array_equal(da[n // 2:].cpu().numpy(), ((a + b) + b)[n // 2:])


# ------------------------------------------------------------------ 4. coverage of the benchmark's kernel symbols
def _norm(name):
  return name.replace('void ', '').replace(' ', '')


def test_every_kernel_symbol_of_the_benchmark_step_is_parity_checked():
  """Kernel symbols of the batch-128 step (debug launch log) and of the newest committed rocprofv3 kernel statistics of
  bench.py (profiles/CURRENT names the file that belongs to this tree) must all have been launched by an
  oracle-checked test: the entry-point cases above or the 2-image 640x640 step."""
  big = _step('bf16', BIG_N)
  checked = dict(COVERED)
  for k, v in _step('bf16', 2).kernels.items():
    checked[k] = checked.get(k, 0) + v
  have = {_norm(k) for k in checked}
  missing = sorted(k for k in big.kernels if _norm(k) not in have)
  print('batch-128 step: %d kernel symbols, %d launches; oracle-checked symbols: %d' % (
      len(big.kernels), sum(big.kernels.values()), len(have)))
  _STEPS.clear()
  torch.cuda.empty_cache()
  assert not missing, 'kernel symbols of the benchmark step that no parity test launches: %s' % missing
  cur = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'CURRENT')
  if not os.path.exists(cur):
    pytest.skip('profiles/CURRENT not present: no rocprof statistics tied to this tree')
  csv_path = os.path.join(os.path.dirname(cur), open(cur).read().strip())
  import csv
  names = [row['Name'] for row in csv.DictReader(open(csv_path))]
  ours = [n for n in names if '::k_' in n or n.startswith('k_') or n.startswith('void k_')]
  assert len(ours) >= 20, 'no library kernels in %s' % csv_path
  missing = sorted(n for n in ours if _norm(n) not in have)
  assert not missing, 'kernels of %s that no parity test launches: %s' % (os.path.basename(csv_path), missing)
