"""-m gpu: parity of the bf16 paths AT THE SIZES OF BASELINE.json configs[1] AND configs[4].

bench.py times two more configurations next to the headline one (`other_configs`): EfficientNetV2-S 224x224 batch 256
forward (configs[1]) and the per-GPU leg of EfficientDet-D7x 1536x1536 batch 8 (configs[4]).  This module puts both
under the discipline tests/test_gpu_bench_shapes.py applies to the headline configuration:

  1. every C-ABI entry point of the path at the configuration's real layer shapes, bf16, against the CPU oracle (the same
     test bodies as tests/test_gpu_kernels.py / tests/test_effnetv2.py): the dense 3x3 convolutions of the Fused-MBConv
     stages at 112x112 / 56x56 / 28x28, the MBConv stages down to 7x7x1536, the D7x stem at 768x768x64, the 384-channel
     BiFPN layers at 192x192, the 768x768x192 expansion (the largest tensor of the D7x step, 1.8 GB at batch 8);
  2. the whole network at a batch the CPU oracle finishes in about a minute (V2-S: 2 images, D7x: 1 image), against the
     oracle that rounds where the engine stores -- LAYER BY LAYER (teacher forced: every stored tensor from the device's
     own stored inputs, tolerance 1.5 bf16 ulp) and END TO END (bounded by the measured conditioning of the map, see
     tests/test_effnetv2.py::test_v2_bf16_end_to_end_conditioning);
  3. the timed batch through a size-independent property: the inference forward of B copies is BIT-IDENTICAL to the
     oracle-checked small batch (no atomics on the inference path, batch-independent reduction order), the D7x train
     step at batch 8 reproduces the losses / gradient direction of the 1-image step it is made of;
  4. coverage: every kernel symbol the timed batch launches is also launched by an oracle-checked test of this module.

The D7x oracle runs cost ~1 minute of host time each (815 GFLOP forward in fp32 on the CPU), which is why this module
keeps them to three.
"""
import os

import numpy as np
import pytest
import torch

from automl_amd import _lib, effnetv2_model, hparams_config, train_lib
from oracle import efficientdet_oracle as orc
from tests import gpu_util as gu
from tests import test_effnetv2 as tv2
from tests import test_gpu_kernels as tk
from tests.test_gpu_bench_shapes import ENGINE_WS_MIB, expand_block_entry_points_equal_their_tiles
from tests.test_gpu_network import _seg_index, drop_scales, make_labels, perturbed_params, rel_err

pytestmark = pytest.mark.gpu

BF16 = gu.DTYPES[1]
TOL_LAYER, TOL_D7X_E2E = 1.2e-2, 1.5e-2      # the tolerances of tests/test_gpu_network.py (TOL_LAYER, TOL_BF16_VS_EMU)
TOL_LAYER_GRAD = 3e-2                        # tests/test_gpu_bench_shapes.py TOL['layer_grad'] / ['layer_wgrad']
COVERED = {}


@pytest.fixture(autouse=True)
def _log_kernel_symbols():
  _lib.launch_log_start()
  yield
  for k, v in _lib.launch_log_stop().items():
    COVERED[k] = COVERED.get(k, 0) + v


def _norm(name):
  return name.replace('void ', '').replace(' ', '')


def _assert_covered(kernels, what):
  have = {_norm(k) for k in COVERED}
  missing = sorted(k for k in kernels if _norm(k) not in have)
  print('%s: %d kernel symbols, %d launches; oracle-checked symbols of this module: %d' % (
      what, len(kernels), sum(kernels.values()), len(have)))
  assert not missing, 'kernel symbols of %s that no parity test of this module launches: %s' % (what, missing)


# =============================================================================== configs[1]: EfficientNetV2-S 224x224
# effnetv2_configs.v2_s_block at 224x224 (efficientnetv2/effnetv2_configs.py:150-157, effnetv2_model.py:313-406):
V2S_CONV = [   # dense 3x3 convolutions of the Fused-MBConv stages: (H = W of the input, cin, cout, stride)
    (112, 24, 24, 1), (112, 24, 96, 2), (56, 48, 192, 1), (56, 48, 192, 2), (28, 64, 256, 1),
]
V2S_PW = [     # 1x1 convolutions: (H = W, cin, cout, view of the input)
    (56, 96, 48, 'bn_swish'), (56, 192, 48, 'bn_swish'), (28, 192, 64, 'bn_swish'), (28, 256, 64, 'bn_swish'),
    (28, 64, 256, 'plain'), (14, 256, 128, 'gate'), (14, 128, 512, 'plain'), (14, 512, 128, 'gate'),
    (14, 128, 768, 'plain'), (14, 768, 160, 'gate'), (14, 160, 960, 'plain'), (14, 960, 160, 'gate'),
    (7, 960, 256, 'gate'), (7, 256, 1536, 'plain'), (7, 1536, 256, 'gate'), (7, 256, 1280, 'plain'),
]
V2S_DW = [(28, 256, 3, 2), (14, 512, 3, 1), (14, 768, 3, 1), (14, 960, 3, 1), (14, 960, 3, 2), (7, 1536, 3, 1)]
N_IMG = 2


def _conv_id(l):
  return '%dx%dx%d->%d_s%d' % (l[0], l[0], l[1], l[2], l[3])


def _pw_id(l):
  return '%dx%dx%d->%d' % (l[0], l[0], l[1], l[2])


def _dw_id(l):
  return '%dx%dx%d_k%ds%d' % (l[0], l[0], l[1], l[2], l[3])


@pytest.mark.parametrize('layer', V2S_CONV, ids=_conv_id)
def test_conv_fwd_at_v2s_224_shapes(layer):
  h, cin, cout, s = layer
  tv2.test_conv_fwd(BF16, (N_IMG, h, h, cin, cout), (3, s), 'plain')        # block inputs are stored tensors


@pytest.mark.parametrize('layer', V2S_CONV, ids=_conv_id)
def test_conv_bwd_at_v2s_224_shapes(layer):
  """edet_conv_bwd_data / edet_conv_bwd_weight (EffNetV2Model.backward): plain dy, and dy with the BatchNorm backward on
  load + accumulation into an already written input gradient (the residual blocks)."""
  h, cin, cout, s = layer
  tv2.test_conv_bwd(BF16, (N_IMG, h, h, cin, cout), (3, s), 'plain')
  tv2.test_conv_bwd(BF16, (N_IMG, h, h, cin, cout), (3, s), 'bn_dy_beta')


@pytest.mark.parametrize('layer', V2S_PW, ids=_pw_id)
def test_pw_at_v2s_224_shapes(layer, monkeypatch):
  h, cin, cout, view = layer
  shape = (N_IMG, h, h, cin, cout)
  tk.test_pw_fwd(BF16, shape, {'gate': 'bn_swish_gate'}.get(view, view), 'auto')
  if cin <= 256:
    # the library picks tiled / streaming by the row count: 2 images of the 56 / 28-row maps run the tiled kernel where
    # the batch-256 forward streams (its instantiation depends on the channel counts only) -- run that one too
    monkeypatch.setenv('EDET_PW_IMPL', 'stream')
    tk.test_pw_fwd(BF16, shape, {'gate': 'bn_swish_gate'}.get(view, view), 'auto')
    monkeypatch.delenv('EDET_PW_IMPL')
  tk.test_pw_bwd_data(BF16, shape, {'bn_swish': 'bn_swish_stats'}.get(view, view), True, 'auto', one_call=True,
                      ws_mib=ENGINE_WS_MIB)


@pytest.mark.parametrize('layer', V2S_DW, ids=_dw_id)
def test_dw_at_v2s_224_shapes(layer):
  h, c, k, s = layer
  tk.test_dw_fwd(BF16, (N_IMG, h, h, c), (k, s), 'bn_swish')
  tk.test_dw_bwd(BF16, (N_IMG, h, h, c), (k, s), 'bn_swish_stats', 'one_call', ws_mib=ENGINE_WS_MIB)


def test_stem_and_se_at_v2s_224_shapes():
  tk.test_stem(BF16, (N_IMG, 224, 224, 24))
  for shape in ((N_IMG, 14, 14, 256, 16), (N_IMG, 14, 14, 512, 32), (N_IMG, 14, 14, 960, 40), (N_IMG, 7, 7, 1536, 64)):
    tk.test_squeeze_excite(BF16, shape)


V2S = 'efficientnetv2-s'
V2S_OVER = 'survival_prob=0,dropout_rate=0'        # the bench's model: 1000 classes, inference
_V2S = {}


def _v2s_small(residual_gamma):
  """The oracle-checked 2-image forward of efficientnetv2-s at 224x224 (inference BatchNorm), kept for the tile test."""
  if residual_gamma not in _V2S:
    over, vals, images = tv2._v2_problem(V2S, 224, N_IMG, False, over=V2S_OVER, bf16=True, residual_gamma=residual_gamma)
    net = effnetv2_model.EffNetV2Model(V2S, over, dtype='bf16', params=vals)
    _lib.launch_log_start()
    got = tv2._v2_device(net, images, False)
    kernels = _lib.launch_log_stop()
    _lib.launch_log_start()
    _V2S[residual_gamma] = (over, vals, images, net, got, kernels)
  return _V2S[residual_gamma]


def test_v2s_224_batch2_forward_layer_by_layer():
  """configs[1] at two images, bf16, random weights with BatchNorm scales ~1: every stored tensor of the forward pass
  (2 + 2 per Fused block + 3 per MBConv block + ... = >= 100 tensors) within TOL_LAYER of the emulating oracle's value
  computed from the device's own stored inputs of that layer; pooled features and the 1000 logits from the device's
  stored head convolution."""
  over, vals, images, net, got, _ = _v2s_small(1.0)
  o = tv2._v2_oracle(V2S, over, vals, 'bf16')
  hook = o.hook = gu.TeacherForce(net.engine)
  with torch.no_grad():
    want = o.forward(images, False)
  print('v2-s 224 B=2 teacher-forced: %d tensors, worst %s' % (len(hook.fwd_err), hook.worst(hook.fwd_err)))
  unstored = {s + ':exp' for s in net.engine.fused_heads}      # (inference, fused MBConv heads: none in V2-S, cin >= 48)
  assert len(hook.fwd_err) >= 100 and set(hook.missing) <= unstored, (len(hook.fwd_err), hook.missing[:8])
  assert max(hook.fwd_err.values()) <= TOL_LAYER, hook.worst(hook.fwd_err, 6)
  tail = {nm: rel_err(got[nm], want[nm]) for nm in ('pooled_features', 'head')}
  assert tuple(got['head'].shape) == (N_IMG, 1000) and max(tail.values()) <= TOL_LAYER, tail


def test_v2s_224_batch2_forward_end_to_end():
  """The same forward end to end against the emulating oracle on the conditioned problem (residual BatchNorm scales
  RESIDUAL_GAMMA), every endpoint within TOL_E2E_BF16; fp32 storage within 1e-3 of the fp32 oracle (north_star)."""
  over, vals, images, net, got, _ = _v2s_small(tv2.RESIDUAL_GAMMA)
  with torch.no_grad():
    emu = tv2._v2_oracle(V2S, over, vals, 'bf16').forward(images, False)
    f32 = tv2._v2_oracle(V2S, over, vals).forward(images, False)
  errs = tv2._v2_errs(got, emu)
  print('v2-s 224 B=2 bf16 end to end vs emulating oracle %s\n   vs fp32 oracle %s' % (
      tv2._fmt(errs), tv2._fmt(tv2._v2_errs(got, f32))))
  bad = {k: v for k, v in errs.items() if v > tv2.TOL_E2E_BF16[k]}
  assert not bad, (bad, errs)
  net32 = effnetv2_model.EffNetV2Model(V2S, over, dtype='f32', params=vals)
  e32 = tv2._v2_errs(tv2._v2_device(net32, images, False), f32)
  print('v2-s 224 B=2 fp32 storage vs fp32 oracle %s' % tv2._fmt(e32))
  assert max(e32.values()) <= 1e-3, e32


V2S_BATCH = 256


def test_v2s_224_batch256_forward_is_bit_identical_to_its_2_image_tiles_and_covered():
  """BASELINE configs[1] at the timed size: 128 copies of the two oracle-checked images.  Inference forward, images are
  independent and nothing on the path depends on the batch (no atomics, SE pooling in batch-independent chunks): every
  copy must give the 2-image endpoints BIT FOR BIT.  The kernel symbols of the batch-256 forward must all have been
  launched by the oracle-checked tests above."""
  over, vals, images, net, small, small_kernels = _v2s_small(1.0)
  big_net = effnetv2_model.EffNetV2Model(V2S, over, dtype='bf16', params=vals)
  reps = V2S_BATCH // N_IMG
  _lib.launch_log_start()
  big = tv2._v2_device(big_net, images.repeat(reps, 1, 1, 1), False)
  kernels = _lib.launch_log_stop()
  _lib.launch_log_start()
  for nm in tv2.V2_ENDPOINTS:
    v = big[nm].view((reps,) + tuple(small[nm].shape))
    for k in (0, 1, reps // 2, reps - 1):
      assert torch.equal(v[k], small[nm]), '%s: copy %d of %d differs from the 2-image forward' % (nm, k, reps)
  for k, v in small_kernels.items():
    COVERED[k] = COVERED.get(k, 0) + v
  _assert_covered(kernels, 'the efficientnetv2-s 224x224 batch-256 forward')
  _V2S.clear()
  del big_net
  torch.cuda.empty_cache()


# =============================================================================== configs[4]: EfficientDet-D7x 1536x1536
D7X = 'efficientdet-d7x'
D7X_SIZE = 1536


def test_entry_points_at_d7x_1536_shapes():
  """efficientnet-b7 stem at 768x768x64, the first expansion 768x768x32 -> 192 with its stride-2 depthwise layer, the
  384-filter BiFPN (hparams_config.py:377-388) separable convolution and 'sum' fusion at 192x192 (level 3), the class
  predictor 192x192x384 -> 810; one image, bf16, against the oracle."""
  tk.test_stem(BF16, (1, D7X_SIZE, D7X_SIZE, 64))
  shape = (1, 768, 768, 32, 192)
  tk.test_pw_fwd(BF16, shape, 'plain', 'auto')
  tk.test_pw_bwd_data(BF16, shape, 'plain', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
  tk.test_dw_fwd(BF16, (1, 768, 768, 192), (3, 2), 'bn_swish')
  tk.test_dw_bwd(BF16, (1, 768, 768, 192), (3, 2), 'bn_swish_stats', 'one_call', ws_mib=ENGINE_WS_MIB)
  shape = (1, 192, 192, 384, 384)
  tk.test_pw_fwd(BF16, shape, 'plain', 'auto')
  tk.test_pw_bwd_data(BF16, shape, 'plain', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
  tk.test_pw_bwd_data(BF16, shape, 'bn_swish_stats', True, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
  tk.test_dw_fwd(BF16, (1, 192, 192, 384), (3, 1), 'plain')
  tk.test_dw_bwd(BF16, (1, 192, 192, 384), (3, 1), 'plain_beta', 'one_call', ws_mib=ENGINE_WS_MIB)
  tk.test_fuse(BF16, ((1, 192, 192, 384), [(_lib.RS_IDENTITY, 192, 192), (_lib.RS_UP2, 96, 96)], 'sum'))
  tk.test_fuse(BF16, ((1, 96, 96, 384), [(_lib.RS_IDENTITY, 96, 96), (_lib.RS_IDENTITY, 96, 96), (_lib.RS_POOL, 192, 192)],
                      'sum'))
  tk.test_pw_fwd(BF16, (1, 192, 192, 384, 810), 'plain', 'auto')
  tk.test_pw_bwd_data(BF16, (1, 192, 192, 384, 810), 'plain', False, 'auto', one_call=True, ws_mib=ENGINE_WS_MIB)
  tk.test_squeeze_excite(BF16, (1, 384, 384, 192, 8))
  tk.test_squeeze_excite(BF16, (1, 48, 48, 3840, 160))


def test_largest_d7x_tensor_entry_points_equal_their_tiles_at_batch_8():
  """768x768x192 at batch 8 is the largest tensor of the D7x step: 906 M elements, 1.81 GB in bf16 (no tensor of this
  configuration passes 2^31 bytes; its offsets pass 2^30 elements).  Every entry point that touches it, on 8 copies of
  the 1-image problem checked against the oracle above: per-image outputs bit-identical in every copy, reductions 8 x."""
  expand_block_entry_points_equal_their_tiles(768, 32, 192, 1, 8, beyond_2gib=False)


def _d7x_problem():
  config = hparams_config.get_efficientdet_config(D7X)
  config.override('image_size=%d' % D7X_SIZE)          # the model's own default (hparams_config.py:377-388)
  vals = perturbed_params(config, 3)
  rng = np.random.default_rng(31)
  images = torch.from_numpy(rng.standard_normal((1, D7X_SIZE, D7X_SIZE, 3)).astype(np.float32)).to(torch.bfloat16).float()
  return config, vals, images


def _d7x_oracle(config, vals, storage='bf16'):
  return orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()}, storage=storage)


_D7X = {}


def _d7x_inference():
  if 'inf' not in _D7X:
    config, vals, images = _d7x_problem()
    net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals)
    _lib.launch_log_start()
    cls, box = net(images, training=False)
    torch.cuda.synchronize()
    kernels = _lib.launch_log_stop()
    _lib.launch_log_start()
    _D7X['inf'] = (config, vals, images, net, [t.float().cpu() for t in cls + box], kernels)
  return _D7X['inf']


def test_d7x_1536_batch1_inference_forward_equals_the_emulating_oracle():
  """BASELINE configs[4] at one image, bf16, inference BatchNorm: 442,260 anchors on levels 3-8; layer by layer (every
  stored tensor of the b7 backbone, the 8 BiFPN cells and the 5-layer towers from the device's own inputs: TOL_LAYER) and
  end to end against the emulating oracle (TOL_D7X_E2E of each level's range, the tolerance tests/test_gpu_network.py
  holds the 256-pixel D7x to)."""
  config, vals, images, net, got, _ = _d7x_inference()
  sizes = [192, 96, 48, 24, 12, 6]
  assert [tuple(t.shape) for t in got[:6]] == [(1, s, s, 810) for s in sizes]
  assert sum(s * s * 9 for s in sizes) == 442260
  with torch.no_grad():
    cref, bref = _d7x_oracle(config, vals).forward(images, False)
  e2e = [round(rel_err(g, w), 5) for g, w in zip(got, cref + bref)]
  print('d7x 1536 B=1 inference bf16 end to end vs emulating oracle: %s' % (e2e,))
  o = _d7x_oracle(config, vals)
  hook = o.hook = gu.TeacherForce(net.engine)
  with torch.no_grad():
    ctf, btf = o.forward(images, False)
  tail = [round(rel_err(g, w), 5) for g, w in zip(got, ctf + btf)]
  print('d7x 1536 B=1 teacher-forced: %d tensors, worst %s; outputs %s' % (
      len(hook.fwd_err), hook.worst(hook.fwd_err), tail))
  assert len(hook.fwd_err) >= 550, (len(hook.fwd_err), hook.missing[:6])
  assert max(hook.fwd_err.values()) <= TOL_LAYER, hook.worst(hook.fwd_err, 6)
  assert max(tail) <= TOL_LAYER, tail
  assert max(e2e) <= TOL_D7X_E2E, e2e


def test_d7x_1536_batch1_training_forward_layer_by_layer():
  """Training-mode BatchNorm (the mode configs[4] is timed in) with stochastic depth, one image of 1536x1536 -- the
  smallest size at which the top pyramid level (6x6) still gives every BatchNorm layer >= 36 samples: every stored
  tensor against the emulating oracle fed the device's own stored inputs and stochastic-depth draws."""
  config, vals, images = _d7x_problem()
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals)
  cls, box = net(images, training=True)
  torch.cuda.synchronize()
  o = _d7x_oracle(config, vals)
  o.drop_scale = drop_scales(net.engine)
  assert o.drop_scale
  hook = o.hook = gu.TeacherForce(net.engine)
  with torch.no_grad():
    ctf, btf = o.forward(images, True)
  tail = [round(rel_err(g, w), 5) for g, w in zip(list(cls) + list(box), ctf + btf)]
  print('d7x 1536 B=1 training forward teacher-forced: %d tensors, worst %s; outputs %s' % (
      len(hook.fwd_err), hook.worst(hook.fwd_err), tail))
  assert len(hook.fwd_err) >= 550, (len(hook.fwd_err), hook.missing[:6])
  assert max(hook.fwd_err.values()) <= TOL_LAYER, hook.worst(hook.fwd_err, 6)
  assert max(tail) <= TOL_LAYER, tail
  net._engines.clear()
  del net
  torch.cuda.empty_cache()


D7X_BATCH = 8


def test_d7x_1536_batch8_inference_forward_is_bit_identical_to_its_1_image_tiles():
  config, vals, images, net, small, _ = _d7x_inference()
  cls, box = net(images.repeat(D7X_BATCH, 1, 1, 1), training=False)
  torch.cuda.synchronize()
  for lvl, (big, one) in enumerate(zip(list(cls) + list(box), small)):
    b = big.float().cpu()
    for k in range(D7X_BATCH):
      assert torch.equal(b[k:k + 1], one), 'output %d: copy %d differs from the 1-image forward' % (lvl, k)
  net._engines.clear()
  torch.cuda.empty_cache()


class _D7xStep(object):
  """One full train step (forward, focal + Huber loss, backward, L2, clipping, SGD / EMA) of d7x at 1536x1536 on `batch`
  copies of one image, stochastic depth on as in the timed configuration but with ONE draw per block shared by all
  copies (per-image draws would make the copies different problems); the generator is seeded identically for every
  batch size, so the 1-image and the 8-image step drop the same blocks."""

  def __init__(self, batch, keep_engine=False):
    config, vals, images = _d7x_problem()
    self.config, self.vals, self.images = config, vals, images
    self.labels1 = make_labels(config, 1, D7X_SIZE, 37)
    labels = {k: np.tile(v, (batch,) + (1,) * (v.ndim - 1)) for k, v in self.labels1.items()}
    labels['normalizer'] = batch * (float(self.labels1['mean_num_positives'].sum()) + 1.0)
    net = self.net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals)
    eng = self.eng = net._ensure_engine(batch, D7X_SIZE, D7X_SIZE)

    def one_draw_per_block():
      for mask, p in eng.drop_masks.values():
        u = torch.rand(1, 1, device=eng.device, generator=eng._rng)
        mask.copy_(((u + p).floor() / p).expand_as(mask))
    eng.refresh_drop_masks = one_draw_per_block
    _lib.launch_log_start()
    eng.forward(net._to_device_images(images.repeat(batch, 1, 1, 1), eng), training=True)
    eng.loss_backward(net._labels_to_device(labels, eng))
    eng.optimizer_step(0.02, 0.9)
    torch.cuda.synchronize()
    self.kernels = _lib.launch_log_stop()
    _lib.launch_log_start()
    self.losses = eng.loss_values()
    # the gradient of the logits (image 0; per-image loss terms are 1 / batch of the 1-image problem's) and the raw
    # (unclipped) gradient of every variable
    self.dlogits = [(v.raw.grad[0, ..., :v.raw.c].float().cpu() * batch) for v in eng.cls_views + eng.box_views]
    self.grads = {name: eng.grad(name).detach().cpu().double().reshape(-1) for name in eng.seg_names}
    if not keep_engine:
      self.release()

  def release(self):
    if self.net is not None:
      self.net._engines.clear()
      self.net = self.eng = None
      torch.cuda.empty_cache()


def _cos(a, b):
  return float((a * b).sum()) / max(float(torch.sqrt((a * a).sum() * (b * b).sum())), 1e-300)


def test_d7x_1536_batch8_train_step_tracks_the_1_image_step_and_is_covered():
  """BASELINE configs[4] per GPU at the timed size (8 images, ~90 GB of activations): 8 copies of one image with the
  loss normalizer scaled by 8 define the same optimisation step as the single image.  What CAN be compared: the forward
  (all six loss values to 2e-3; measured 3e-5), the gradient of the logits (a pointwise function of the forward outputs:
  2e-2 of its max for the class logits) and the variable gradients ONE layer behind the class loss (the class predict
  layer: cosine >= 0.8; measured 0.90, against 0.45 in the towers, 0.05 in the BiFPN and 0.02 in the backbone).  Deeper into the backward pass the bf16 train step of this network is chaotic in the ORACLE ITSELF: the cosine
  between the oracle's own clipped gradient of efficientdet-d7x and the one it computes after 0.05 % of the input pixels
  moved by one bf16 ulp is 0.04 with bf16 storage (0.96 with fp32 storage; tests/test_oracle_conditioning.py) -- 8
  BiFPN cells and 55 blocks of batch-statistics BatchNorm on random weights -- so no implementation can reproduce a
  direction there; the per-depth cosines are printed.  The backward kernels at this size are pinned layer by layer
  (next test) and through the bit-identical tiles above.  Every kernel symbol of the batch-8 step must be launched by
  an oracle-checked test of this module."""
  one, big = _D7xStep(1), _D7xStep(D7X_BATCH)
  for k in ('cls_loss', 'box_loss', 'det_loss', 'reg_l2_loss', 'loss'):
    assert abs(big.losses[k] - one.losses[k]) <= 2e-3 * abs(one.losses[k]) + 1e-6, (k, big.losses[k], one.losses[k])
  derr = [round(rel_err(b, o), 4) for b, o in zip(big.dlogits, one.dlogits)]
  nl = len(derr) // 2
  print('d7x 1536 batch 8 vs 1, gradient of the logits per level: class %s, box %s' % (derr[:nl], derr[nl:]))
  # class logits: a smooth function of logits that agree to a bf16 ulp (measured 7.4e-3).  Box outputs: the Huber
  # gradient saturates at +-delta (0.1) and the zero-bias box outputs of the training-mode network move by tenths of
  # their range between two runs (the measured-conditioning bound of tests/test_gpu_bench_shapes.py), so signs flip: printed, not held
  assert max(derr[:nl]) <= 2e-2, 'gradient of the class logits, batch 8 vs 1: %s' % (derr,)
  groups = {'predict': [], 'tower': [], 'fpn': [], 'backbone': []}
  for n in one.grads:
    if one.grads[n].numel() < 2:
      continue
    g = 'predict' if n.startswith('class_net/class-predict/') else ('tower' if n.startswith(('class_net/', 'box_net/')) else
                                            ('fpn' if n.startswith(('fpn_cells/', 'resample_p')) else 'backbone'))
    groups[g].append(_cos(big.grads[n] * D7X_BATCH, one.grads[n]))
  profile = {g: (round(float(np.median(v)), 4), round(float(np.min(v)), 4)) for g, v in groups.items()}
  print('d7x 1536 batch 8 vs 1: losses %s vs %s; per-variable gradient cosine (median, min) by depth: %s' % (
      {k: round(v, 4) for k, v in big.losses.items()}, {k: round(v, 4) for k, v in one.losses.items()}, profile))
  assert profile['predict'][1] >= 0.8, profile       # measured 0.89 / 0.90: one layer of bf16 chaos behind the class loss
  for k, v in one.kernels.items():
    COVERED[k] = COVERED.get(k, 0) + v
  if 'inf' in _D7X:
    for k, v in _D7X['inf'][5].items():
      COVERED[k] = COVERED.get(k, 0) + v
  _assert_covered(big.kernels, 'the efficientdet-d7x 1536x1536 batch-8 train step')
  _D7X.clear()


@pytest.mark.skipif(os.environ.get('EDET_SKIP_SLOW') == '1', reason='EDET_SKIP_SLOW=1: ~4 minutes of host autograd')
def test_d7x_1536_batch1_bf16_train_step_layer_by_layer():
  """The bf16 train step of efficientdet-d7x at 1536x1536 (one image, stochastic depth on, the device's draws handed to
  the oracle) against the storage-emulating oracle with teacher forcing, forward AND backward: every stored activation,
  every stored gradient buffer and every variable's gradient against the oracle's value computed from the DEVICE's
  stored inputs of that layer (tests/test_gpu_bench_shapes.py does the same for the headline configuration).  This is
  what pins the backward kernels at the D7x shapes -- 768x768 maps, 3840-channel stages, 384-filter BiFPN, 5-layer
  towers -- where end-to-end gradient comparisons are meaningless (previous test).  ~2.4 TFLOP of fp32 autograd on the
  host: the slowest test of the suite."""
  step = _D7xStep(1, keep_engine=True)
  config, vals, images = step.config, step.vals, step.images
  labels = {k: torch.from_numpy(v) for k, v in step.labels1.items()}
  o = _d7x_oracle(config, vals)
  with torch.no_grad():
    o.forward(images[:1, :64, :64], False)         # registers the trainable list
  o.drop_scale = drop_scales(step.eng)
  hook = o.hook = gu.TeacherForce(step.eng)
  P = o.params()
  names = o.trainable_names()
  for n in names:
    P[n].requires_grad_(True)
  cls, box = o.forward(images, True)
  det, _, _ = orc.detection_loss(config, cls, box, labels)
  l2 = config.weight_decay * sum((P[n]**2).sum() / 2 for n in names if orc.is_l2_regularised(n))
  (det + l2).backward()
  print('d7x 1536 teacher-forced train step: %d activations, worst %s; %d gradient buffers, worst %s' % (
      len(hook.fwd_err), hook.worst(hook.fwd_err), len(hook.bwd_err), hook.worst(hook.bwd_err)))
  assert len(hook.fwd_err) >= 550 and len(hook.bwd_err) >= 500, (len(hook.fwd_err), len(hook.bwd_err), hook.missing[:8])
  assert max(hook.fwd_err.values()) <= TOL_LAYER, hook.worst(hook.fwd_err, 6)
  assert max(hook.bwd_err.values()) <= TOL_LAYER_GRAD, hook.worst(hook.bwd_err, 6)
  werr, bn_mine, bn_ref = {}, [], []
  gmax = max(float(P[n].grad.abs().max()) for n in names if P[n].grad is not None)
  for n in names:
    g = P[n].grad
    if g is None or n.rsplit('/', 1)[-1].startswith('WSM'):
      continue
    mine = step.eng.grad(n).cpu().reshape(g.shape)
    if n.endswith('/bias') and 'predict' not in n and '/se/' not in n:
      assert float(mine.abs().max()) == 0.0, n        # a bias in front of a BatchNorm: analytically zero
      continue
    e = float((mine - g).abs().max()) / max(float(g.abs().max()), 1e-4 * gmax)
    if n.endswith('/gamma') or n.endswith('/beta'):
      # whole-tensor sums that cancel by orders of magnitude: checked together as one vector (as for d0)
      bn_mine.append(mine.reshape(-1).double().numpy())
      bn_ref.append(g.reshape(-1).double().numpy())
      continue
    werr[n] = e
  bn_mine, bn_ref = np.concatenate(bn_mine), np.concatenate(bn_ref)
  bn_err = float(np.linalg.norm(bn_mine - bn_ref) / np.linalg.norm(bn_ref))
  print('d7x 1536 teacher-forced: %d kernel gradients, worst %s; BatchNorm gamma / beta vector error %.4f' % (
      len(werr), gu.TeacherForce.worst(werr, 5), bn_err))
  assert max(werr.values()) <= TOL_LAYER_GRAD, gu.TeacherForce.worst(werr, 8)
  assert bn_err <= 3e-2, bn_err
  step.release()
