"""EffNetV2Model: the reference's EfficientNetV2 interface on MI355X kernels (forward path).

Mirror of ``efficientnetv2/effnetv2_model.py``: ``EffNetV2Model(model_name, model_config, include_top)
(inputs[B,H,W,3], training=False, with_endpoints=False)`` (:532-658) built from Stem (:409-432),
MBConvBlock (:187-310), FusedMBConvBlock (:313-406), SE (:105-147) and Head (:435-497).
Variable names follow the Keras model (model name as prefix, ``blocks_%d``, per-block ``conv2d[_1]`` /
``tpu_batch_normalization[_1,_2]`` counters, ``se/conv2d[_1]``, ``head/...``).

Scope: the forward pass in both BatchNorm modes and the backward
pass of the whole graph (``backward(d_outputs)``: Fused-MBConv dense convolutions, MBConv, SE, stochastic depth,
head, dense layer).  The classifier's loss / optimizer / dropout are outside the detection hot path:
``training=True`` refuses dropout (pass ``model_config='dropout_rate=0'``).
"""
import collections
import ctypes
import math

import numpy as np
import torch

from automl_amd import _lib
from automl_amd import effnetv2_configs
from automl_amd import engine as engine_lib
from automl_amd import netspec as netspec_lib
from automl_amd import utils
from automl_amd._lib import ACT_NONE, ACT_SWISH, call, ptr
from automl_amd.netspec import ParamSpec


class V2Spec(object):
  """Structure + variable list of one EffNetV2Model (creation order == the Keras layer order)."""

  def __init__(self, mconfig, include_top=True):
    self.mconfig = mconfig
    self.include_top = include_top
    if mconfig.act_fn not in ('silu', 'swish', None):
      raise ValueError('act_fn %r is out of scope (silu/swish only)' % (mconfig.act_fn,))
    if mconfig.bn_type not in (None, 'tpu_bn'):
      raise ValueError('bn_type %r is out of scope (batch norm only)' % (mconfig.bn_type,))
    if mconfig.data_format != 'channels_last':
      raise ValueError('only channels_last (NHWC) is built')
    self.bn_momentum = float(mconfig.bn_momentum)
    self.bn_epsilon = float(mconfig.bn_epsilon)
    self.name = mconfig.model_name
    self.stem_filters, self.blocks = effnetv2_configs.expand_blocks(mconfig)
    # stochastic depth (effnetv2_model.py:624-629): survival_prob 0.8 -> per block 1 - 0.2 * idx / n
    sp = mconfig.survival_prob
    self.survival_probs = [(1.0 - (1.0 - sp) * float(b.index) / len(self.blocks)) if sp else None
                           for b in self.blocks]
    self.head_filters = effnetv2_configs.round_filters(mconfig.feature_size or 1280, mconfig)
    self.num_classes = mconfig.num_classes if include_top else 0
    self.params = []
    self._build()

  def _add(self, name, shape, init, trainable=True):
    self.params.append(ParamSpec(name, tuple(shape), init, trainable))

  def _bn(self, scope, c):
    self._add(scope + '/gamma', (c,), 'ones')
    self._add(scope + '/beta', (c,), 'zeros')
    self._add(scope + '/moving_mean', (c,), 'zeros', False)
    self._add(scope + '/moving_variance', (c,), 'ones', False)

  def _build(self):
    n = self.name
    self._add(n + '/stem/conv2d/kernel', (3, 3, 3, self.stem_filters), 'conv')
    self._bn(n + '/stem/tpu_batch_normalization', self.stem_filters)
    bn_names = ['tpu_batch_normalization', 'tpu_batch_normalization_1', 'tpu_batch_normalization_2']
    for b in self.blocks:
      s = '%s/blocks_%d' % (n, b.index)
      cexp = b.input_filters * b.expand_ratio
      k = b.kernel_size
      if b.conv_type == 0:       # MBConv
        ci = bi = 0
        if b.expand_ratio != 1:
          self._add(s + '/conv2d/kernel', (1, 1, b.input_filters, cexp), 'conv')
          self._bn('%s/%s' % (s, bn_names[bi]), cexp)
          ci, bi = 1, 1
        self._add(s + '/depthwise_conv2d/depthwise_kernel', (k, k, cexp, 1), 'conv')
        self._bn('%s/%s' % (s, bn_names[bi]), cexp)
        bi += 1
        if b.se_filters:
          self._se(s, cexp, b.se_filters)
        self._add('%s/%s/kernel' % (s, 'conv2d_1' if ci else 'conv2d'), (1, 1, cexp, b.output_filters), 'conv')
        self._bn('%s/%s' % (s, bn_names[bi]), b.output_filters)
      elif b.conv_type == 1:     # Fused-MBConv
        if b.expand_ratio != 1:
          self._add(s + '/conv2d/kernel', (k, k, b.input_filters, cexp), 'conv')
          self._bn(s + '/tpu_batch_normalization', cexp)
          if b.se_filters:
            self._se(s, cexp, b.se_filters)
          self._add(s + '/conv2d_1/kernel', (1, 1, cexp, b.output_filters), 'conv')
          self._bn(s + '/tpu_batch_normalization_1', b.output_filters)
        else:
          if b.se_filters:
            self._se(s, cexp, b.se_filters)
          self._add(s + '/conv2d/kernel', (k, k, cexp, b.output_filters), 'conv')
          self._bn(s + '/tpu_batch_normalization', b.output_filters)
      else:
        raise ValueError('conv_type %r is out of scope' % (b.conv_type,))
    self._add(n + '/head/conv2d/kernel', (1, 1, self.blocks[-1].output_filters, self.head_filters), 'conv')
    self._bn(n + '/head/tpu_batch_normalization', self.head_filters)
    if self.num_classes:
      self._add(n + '/head/dense/kernel', (self.head_filters, self.num_classes), 'dense')
      self._add(n + '/head/dense/bias', (self.num_classes,), 'zeros')

  def _se(self, s, c, se):
    self._add(s + '/se/conv2d/kernel', (1, 1, c, se), 'conv')
    self._add(s + '/se/conv2d/bias', (se,), 'zeros')
    self._add(s + '/se/conv2d_1/kernel', (1, 1, se, c), 'conv')
    self._add(s + '/se/conv2d_1/bias', (c,), 'zeros')

  def count_params(self):
    """All weights incl. BatchNorm moving statistics (Keras model.count_params(),
    effnetv2_model_test.py:24-52)."""
    return sum(int(np.prod(p.shape)) if p.shape else 1 for p in self.params)

  def reduction_indices(self):
    """Blocks whose outputs are reduction_1..5 (effnetv2_model.py:618-622)."""
    b = self.blocks
    return [i for i in range(len(b)) if i == len(b) - 1 or b[i + 1].stride > 1]


def init_value(spec, rng):
  """conv: N(0, sqrt(2/(kh*kw*cout))) (effnetv2_model.py:40-61); dense: U(+-1/sqrt(cout)) (:64-81)."""
  if spec.init == 'dense':
    lim = 1.0 / math.sqrt(spec.shape[1])
    return rng.uniform(-lim, lim, spec.shape).astype(np.float32)
  return netspec_lib.init_value(spec, rng)


def init_params(spec, seed=0):
  rng = np.random.default_rng(seed)
  return collections.OrderedDict((p.name, init_value(p, rng)) for p in spec.params)


class V2Engine(engine_lib.Engine):
  """Launch plan of one EffNetV2Model on one MI355X (buffers, BatchNorm plumbing and the layer
  primitives come from engine.Engine)."""

  def __init__(self, spec, batch_size, image_size, dtype='bf16', device='cuda:0', seed=0, params=None, arena=None):
    if arena is None and (params is None or any(p.name not in params for p in spec.params)):
      params = {**init_params(spec, seed), **(params or {})}      # a partial set keeps the other initial values
    super().__init__(spec.mconfig, batch_size, image_size, dtype=dtype, device=device, seed=seed,
                     params=params, spec=spec, arena=arena)

  def forward(self, images, training=False, update_moving=True):
    """images: device tensor [B,H,W,3] in the engine dtype.  Fills self.endpoints / self.outputs."""
    spec = self.spec
    assert tuple(images.shape) == (self.batch, self.image_size[0], self.image_size[1], 3), images.shape
    assert images.dtype == self.tdtype and images.is_contiguous()
    self._begin(training, update_moving)
    self.images = images
    n, h, w = self.batch, self.image_size[0], self.image_size[1]
    name = spec.name
    oh, _, _ = utils.same_padding(h, 3, 2)
    ow, _, _ = utils.same_padding(w, 3, 2)
    y0 = engine_lib.Raw(self, 'stem', n, oh, ow, spec.stem_filters)
    bn0 = self.get_bn(name + '/stem/tpu_batch_normalization', spec.stem_filters)
    call('edet_stem_fwd', ptr(images), n, h, w, ptr(self.param(name + '/stem/conv2d/kernel')), ptr(y0.data),
         spec.stem_filters, y0.ld, ptr(self.partials) if training else None, ctypes.byref(self._nparts),
         self.dtype, self.stream, nbytes=(n * h * w * 3 + y0.rows * y0.c) * self.esize)
    self._bn_forward(bn0, y0.rows, self._nparts.value)
    x = engine_lib.View(y0, bn0, ACT_SWISH)
    if training:
      v0, wstem = x, name + '/stem/conv2d/kernel'

      def stem_bwd():
        g = self._gview(v0)
        # behind the partial sums that earlier layers left for the deferred reductions (Engine._ws / _ws_mark)
        call('edet_stem_bwd_weight', ptr(images), n, h, w, ctypes.byref(g), ptr(self.grad(wstem)), *self._ws(), self.dtype,
             self.stream, nbytes=(n * h * w * 3 + y0.rows * y0.c) * self.esize)
        self._ws_mark()
      self.tape.append(stem_bwd)
    if training or spec.blocks[0].has_residual or spec.blocks[0].conv_type == 1:
      # the dense convolutions (and a first block that adds its input back) read a STORED tensor: the stem
      # output is materialised in activated form (one extra pass over a stride-2, 24..32-channel map)
      x = self.bn_res('stem:out', x, None)
    reds = set(spec.reduction_indices())
    self.endpoints = {'stem': x}
    ridx = 0
    for b in spec.blocks:
      scope = '%s/blocks_%d' % (name, b.index)
      x = self._mbconv(x, b, scope) if b.conv_type == 0 else self._fused_mbconv(x, b, scope)
      self.endpoints['block_%d' % b.index] = x
      if b.index in reds:
        ridx += 1
        self.endpoints['reduction_%d' % ridx] = x
    self.endpoints['features'] = x
    hv = self.pw('head:conv', x, name + '/head/conv2d/kernel', spec.head_filters,
                 bn=name + '/head/tpu_batch_normalization', act=ACT_SWISH)
    self.endpoints['head_1x1'] = hv
    r = hv.raw
    pooled = self.buf('head:pool', (n, r.c), torch.float32)
    call('edet_se_pool', ctypes.byref(hv.tview()), ptr(pooled), ptr(self.partials), self.partials.numel() * 4,
         self.dtype, self.stream, nbytes=r.rows * r.c * self.esize)
    self.pooled_sum, self.pooled_inv_hw = pooled, 1.0 / (r.h * r.w)
    self.head_view = hv
    self.logits = None
    if spec.num_classes:
      pv = engine_lib.Raw(self, 'head:pooled', n, 1, 1, r.c, needs_grad=False)
      call('edet_cast', ptr(pooled), ptr(pv.data), n * r.c, self.dtype, self.stream)
      inv = self.buf('head:inv_hw', (2, r.c), torch.float32)
      inv[0].fill_(self.pooled_inv_hw)
      inv[1].zero_()
      view = engine_lib.View(pv)
      wt, ldk, _, _ = self._pw_copies(name + '/head/dense/kernel', r.c, spec.num_classes)
      out = engine_lib.Raw(self, 'head:logits', n, 1, 1, spec.num_classes, needs_grad=False)
      tv = view.tview()
      tv.scale, tv.shift = inv[0].data_ptr(), inv[1].data_ptr()      # mean = sum / (H*W) folded into the load
      call('edet_pw_fwd', ctypes.byref(tv), ptr(wt), ldk, ptr(self.param(name + '/head/dense/bias')),
           ptr(out.data), spec.num_classes, out.ld, None, ctypes.byref(self._nparts), self.dtype, self.stream)
      self.logits = out
      self._fc = (pv, tv, inv)
    return self.logits

  def backward(self, d_out):
    """Gradients of every variable for a given gradient of the model output (after forward(training=True)):
    d_out = d(logits) [B, num_classes] with include_top, else d(pooled features) [B, feature_size].  Fills
    grads_flat (get_grads()).  The loss itself -- softmax cross-entropy with label smoothing in the reference's
    classifier training, efficientnetv2/main.py -- stays with the caller: classifier training is outside the
    detection hot path, the backward of the Fused-MBConv / MBConv / SE / head graph is what is built here."""
    spec = self.spec
    assert self.training, 'run forward(training=True) first'
    name = spec.name
    hv = self.head_view
    r = hv.raw
    n, c = r.n, r.c
    d_out = torch.as_tensor(d_out).to(device=self.device, dtype=torch.float32).reshape(n, -1)
    if spec.num_classes:
      pv, tv, inv = self._fc
      ncls = spec.num_classes
      dl = engine_lib.Raw(self, 'head:dlogits', n, 1, 1, ncls, needs_grad=False)
      dl.data.zero_()
      dl.data.reshape(n, -1)[:, :ncls] = d_out.to(self.tdtype)
      g = _lib.GView(ptr(dl.data), None, None, None, None, n, 1, 1, ncls, dl.ld)
      wname = name + '/head/dense/kernel'
      _, _, wcopy, ldn = self._pw_copies(wname, c, ncls)
      call('edet_pw_bwd_weight', ctypes.byref(tv), ctypes.byref(g), ptr(self.grad(wname)), *self._ws(), self.dtype, self.stream)
      self._ws_mark()
      self.grad(name + '/head/dense/bias').add_(dl.data.reshape(n, -1)[:, :ncls].float().sum(0))
      dpv = self.buf('head:dpooled', (n, 1, 1, pv.ld), self.tdtype)
      epi = _lib.BwdEpi(ptr(dpv), 0, None, None, None, None)
      plain = _lib.TView(ptr(pv.data), None, None, None, ACT_NONE, n, 1, 1, c, pv.ld)
      call('edet_pw_bwd_data', ctypes.byref(g), ptr(wcopy), ldn, ctypes.byref(plain), ctypes.byref(epi),
           ctypes.byref(self._nparts), self.dtype, self.stream)
      d_mean = dpv.reshape(n, -1)[:, :c].float()          # d(mean-pooled features)
    else:
      d_mean = d_out
    # global average pooling backward: every pixel receives d_mean / (H*W); through swish' and the head BatchNorm
    dpool = self.buf('head:dpool', (n, c), torch.float32)
    dpool.copy_(d_mean * self.pooled_inv_hw)
    gbuf = r.ensure_grad()
    gbuf.zero_()
    ones = self.buf('ones:%d:%d' % (n, c), (n, c), torch.float32)
    ones.fill_(1.0)
    gv = _lib.TView(ptr(r.data), ptr(hv.bn.scale), ptr(hv.bn.shift), ptr(ones), hv.act, r.n, r.h, r.w, c, r.ld)
    call('edet_se_gate_bwd', ctypes.byref(gv), ptr(gbuf), ptr(dpool), ptr(hv.bn.mean), ptr(hv.bn.rstd),
         ptr(self.partials), ctypes.byref(self._nparts), self.dtype, self.stream)
    self._bn_bwd_finalize(hv.bn, self._nparts.value)
    r.grad_written = True
    super().backward()

  def _fused_mbconv(self, xin, b, scope):
    """FusedMBConvBlock.call (effnetv2_model.py:373-406)."""
    cexp = b.input_filters * b.expand_ratio
    x = xin
    if b.expand_ratio != 1:
      x = self.conv(scope + ':exp', x, scope + '/conv2d/kernel', b.kernel_size, b.stride, cexp,
                    bn=scope + '/tpu_batch_normalization', act=ACT_SWISH)
      if b.se_filters:
        x = self.se(scope + ':se', x, scope, b.se_filters)
      y = self.pw(scope + ':proj', x, scope + '/conv2d_1/kernel', b.output_filters,
                  bn=scope + '/tpu_batch_normalization_1', act=ACT_NONE)
    else:
      if b.se_filters:
        raise ValueError('SE in an expand_ratio == 1 fused block gates the block input: out of scope')
      y = self.conv(scope + ':conv', x, scope + '/conv2d/kernel', b.kernel_size, b.stride, b.output_filters,
                    bn=scope + '/tpu_batch_normalization', act=ACT_SWISH)     # act because no expansion
    return self.bn_res(scope + ':out', y, xin if b.has_residual else None,
                       survival_prob=self.spec.survival_probs[b.index])


class EffNetV2Model(object):
  """EfficientNetV2 / EfficientNet (V2 code base) forward model, same call surface as the reference."""

  def __init__(self, model_name='efficientnetv2-s', model_config=None, include_top=True, name=None,
               dtype='bf16', device='cuda:0', seed=0, params=None):
    self.name = name or model_name
    self.cfg_model = effnetv2_configs.model_config(model_name, model_config)
    self._mconfig = self.cfg_model
    self.include_top = include_top
    self.spec = V2Spec(self.cfg_model, include_top)
    self._dtype, self._device, self._seed, self._init_params = dtype, device, seed, params
    self.engine = None
    self.endpoints = None

  def count_params(self):
    return self.spec.count_params()

  def _ensure_engine(self, batch, height, width):
    e = self.engine
    if e is None or e.batch != batch or e.image_size != (height, width):
      # a new shape gets new buffers; the variables stay in the arena the previous executor used
      self.engine = V2Engine(self.spec, batch, (height, width), dtype=self._dtype, device=self._device,
                             seed=self._seed, params=self._init_params, arena=None if e is None else e.arena)
    return self.engine

  def backward(self, d_outputs):
    """Gradients of every variable w.r.t. a given d(outputs) of the last training=True call -> {name: array}."""
    self.engine.backward(d_outputs)
    return self.engine.get_grads()

  def __call__(self, inputs, training=False, with_endpoints=False):
    """-> logits [B,num_classes] (include_top) or pooled features [B,feature_size]; with_endpoints:
    [outputs, reduction_1..5] (effnetv2_model.py:595-658).  Tensors are device torch tensors."""
    if training and (self._mconfig.dropout_rate or self._mconfig.conv_dropout):
      raise ValueError('training=True with dropout is outside the built path; '
                       "override model_config='dropout_rate=0'")
    if isinstance(inputs, np.ndarray):
      inputs = torch.from_numpy(inputs)
    if inputs.dim() != 4 or inputs.shape[-1] != 3:
      raise ValueError('inputs must be [batch, height, width, 3], got %s' % (tuple(inputs.shape),))
    b, h, w = int(inputs.shape[0]), int(inputs.shape[1]), int(inputs.shape[2])
    eng = self._ensure_engine(b, h, w)
    eng.forward(inputs.to(device=eng.device, dtype=eng.tdtype).contiguous(), training=training)
    self.endpoints = {k: self._materialise(eng, v) for k, v in eng.endpoints.items()
                      if k.startswith('reduction_') or k == 'features'}
    pooled = (eng.pooled_sum * eng.pooled_inv_hw)
    self.endpoints['pooled_features'] = pooled
    if eng.logits is not None:
      outputs = eng.logits.data.reshape(b, -1)[:, :self.spec.num_classes].float()
    else:
      outputs = pooled
    self.endpoints['head'] = outputs
    if with_endpoints:
      return [outputs] + [self.endpoints['reduction_%d' % i] for i in range(1, 6)
                          if 'reduction_%d' % i in self.endpoints]
    return outputs

  call = __call__

  @staticmethod
  def _materialise(eng, view):
    """Block outputs are stored plain (edet_bn_res): a strided torch view of the buffer."""
    assert view.bn is None and view.gate is None and view.act == ACT_NONE
    return view.raw.data[..., :view.raw.c]

  def set_weights(self, values):
    if self.engine is None:
      self._init_params = dict(values) if self._init_params is None else {**self._init_params, **values}
    else:
      self.engine.set_params(values)

  def get_weights(self):
    if self.engine is None:
      raise RuntimeError('the network has not been built yet (call it once)')
    return self.engine.get_params()


def get_model(model_name, model_config=None, include_top=True, weights=None, **kwargs):
  """effnetv2_model.get_model (:661-760).  ``weights``: None (random initialisation) or the path of a checkpoint /
  checkpoint directory, as the reference's last branch (``tf.train.latest_checkpoint`` + ``load_weights``); the named
  pretrained sets ('imagenet', 'imagenet21k', ...) are downloads and cannot be fetched offline."""
  model = EffNetV2Model(model_name, model_config, include_top, **kwargs)
  if weights:
    import os
    if weights in ('imagenet', 'imagenet21k', 'imagenet21k-ft1k', 'jft') or not (
        os.path.isdir(weights) or os.path.exists(weights + '.index')):
      raise ValueError('pretrained weights %r cannot be fetched here; pass a checkpoint path or set_weights()' % (weights,))
    from automl_amd import util_keras
    util_keras.restore_ckpt(model, weights, ema_decay=0, skip_mismatch=False)
  return model
