"""CPU oracle of the EfficientNetV2 forward path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/ and bench.py's cpu_baseline leg may import this module.  Plain fp32 PyTorch-CPU restatement
of ``efficientnetv2/effnetv2_model.py`` (paths relative to the reference root): Stem :409-432,
MBConvBlock.call :279-310, FusedMBConvBlock.call :373-406, SE.call :135-147, Head.call :472-497,
EffNetV2Model.call :595-658; activation ``silu`` = x*sigmoid(x) (efficientnetv2/utils.py:27-31);
BatchNorm momentum 0.9, epsilon 1e-3 (efficientnetv2/hparams.py:228-229).

PARITY STATUS: as for oracle/efficientdet_oracle.py -- the arithmetic lives in TensorFlow, which cannot
be installed here; this restatement is pinned by the reference's RNG-free known answers for the path
(the 15 parameter counts of effnetv2_model_test.py:24-52, checked in tests/test_effnetv2.py), by the
direct-loop twin for the dense convolution (oracle/direct_loops.py) and by the TF 'SAME' padding rule
shared with the EfficientDet oracle, and by the outputs + variable inventory of the reference's own EffNetV2Model
code executed on a torch-backed tf.keras stand-in (tests/golden/make_golden_graph_v2.py; wiring, not layer
arithmetic).  Conv-output parity versus the TensorFlow binary is UNPINNED.
"""
import torch

from automl_amd import effnetv2_configs
from oracle import efficientdet_oracle as det

conv2d_same, depthwise_same, swish = det.conv2d_same, det.depthwise_same, det.swish


class V2Oracle(object):
  """forward(images NHWC fp32, training) -> dict of endpoints (NHWC) incl. 'head' (logits or pooled).

  storage='bf16': the same graph with a round-to-nearest-even bfloat16 rounding exactly where the product path
  (automl_amd/effnetv2_model.V2Engine on engine.Engine) stores a tensor or feeds a matrix-core operand in bf16 -- the
  scheme of efficientdet_oracle.Oracle (its q / qg / qop / pw / dw are used as they are): images, every raw convolution
  output (BatchNorm statistics are those of the stored values), the activated operand and the weight copy of every
  stem / dense k x k / pointwise convolution, the materialised stem output and every block output, the pooled sums and
  the logits of the classifier head; in the backward pass the gradient buffers of those tensors.  `hook` = teacher
  forcing (oracle/teacher_force.py): keys are the executor's buffer names ('stem', 'stem:out', '<scope>:exp' / ':conv'
  / ':dw' / ':proj' / ':out', 'head:conv')."""

  q, qg, qop, pw, dw = det.Oracle.q, det.Oracle.qg, det.Oracle.qop, det.Oracle.pw, det.Oracle.dw

  def __init__(self, model_name='efficientnetv2-s', model_config=None, include_top=True, params=None, seed=0,
               storage='f32'):
    assert storage in ('f32', 'bf16'), storage
    self.emulate = storage == 'bf16'
    self.hook = None
    self.mconfig = effnetv2_configs.model_config(model_name, model_config)
    self.include_top = include_top
    self.store = det.ParamStore(params, seed)
    self.new_moving = {}
    # stochastic depth: block scope -> per-image scale floor(p + u) / p ([B] tensor), an INPUT of the oracle
    # (TF's RNG stream cannot be reproduced; the tests pass the draws of the device path).  Empty = off.
    self.drop_scale = {}

  def conv(self, x, w, stride, key):
    """Dense k x k convolution as the implicit-GEMM MFMA kernels compute it: bf16 operands (zero 'SAME' padding
    after the rounding), fp32 accumulate, bf16 store; a BatchNorm always follows."""
    return self.q(conv2d_same(self.qop(x), self.qop(w), stride), key, None)

  def _residual(self, x, inputs, b, scope, training):
    if not b.has_residual:
      return x
    if training and scope in self.drop_scale:
      x = x * self.drop_scale[scope].view(-1, 1, 1, 1)      # utils.drop_connect (efficientnetv2/utils.py:292-307)
    return x + inputs

  def bn(self, x, name, training, grad_key=None):
    """grad_key: the executor's buffer that holds dz, the gradient w.r.t. this BatchNorm's output."""
    c = x.shape[1]
    g = self.store.get(name + '/gamma', (c,), det.ones)
    b = self.store.get(name + '/beta', (c,), det.zeros)
    mm = self.store.get(name + '/moving_mean', (c,), det.zeros, trainable=False)
    mv = self.store.get(name + '/moving_variance', (c,), det.ones, trainable=False)
    eps, mom = self.mconfig.bn_epsilon, self.mconfig.bn_momentum
    if training:
      mean = x.mean(dim=(0, 2, 3))
      var = x.var(dim=(0, 2, 3), unbiased=False)
      n = x.numel() // c
      with torch.no_grad():
        unbiased = var * (n / (n - 1.0)) if n > 1 else var
        self.new_moving[name + '/moving_mean'] = mm * mom + mean * (1 - mom)
        self.new_moving[name + '/moving_variance'] = mv * mom + unbiased * (1 - mom)
    else:
      mean, var = mm, mv
    inv = torch.rsqrt(var + eps) * g
    return self.qg(x * inv.view(1, -1, 1, 1) + (b - mean * inv).view(1, -1, 1, 1), grad_key)

  def se(self, x, scope, c, se_filters):
    P = self.store
    w1 = P.get(scope + '/se/conv2d/kernel', (1, 1, c, se_filters), det.conv_kernel_init)
    b1 = P.get(scope + '/se/conv2d/bias', (se_filters,), det.zeros)
    w2 = P.get(scope + '/se/conv2d_1/kernel', (1, 1, se_filters, c), det.conv_kernel_init)
    b2 = P.get(scope + '/se/conv2d_1/bias', (c,), det.zeros)
    s = x.mean(dim=(2, 3), keepdim=True)
    s = conv2d_same(swish(conv2d_same(s, w1, 1, b1)), w2, 1, b2)
    return self.qg(torch.sigmoid(s) * x)       # the gated gradient D is stored before edet_se_gate_bwd

  def mbconv(self, inputs, b, scope, training):
    P = self.store
    x = inputs
    cexp = b.input_filters * b.expand_ratio
    bn = ['tpu_batch_normalization', 'tpu_batch_normalization_1', 'tpu_batch_normalization_2']
    bi = ci = 0
    if b.expand_ratio != 1:
      w = P.get(scope + '/conv2d/kernel', (1, 1, b.input_filters, cexp), det.conv_kernel_init)
      x = swish(self.bn(self.pw(x, w, key=scope + ':exp'), '%s/%s' % (scope, bn[bi]), training, scope + ':exp#grad'))
      ci, bi = 1, 1
    wd = P.get(scope + '/depthwise_conv2d/depthwise_kernel', (b.kernel_size, b.kernel_size, cexp, 1),
               det.conv_kernel_init)
    x = swish(self.bn(self.dw(x, wd, b.stride, key=scope + ':dw'), '%s/%s' % (scope, bn[bi]), training,
                      scope + ':dw#grad'))
    bi += 1
    if b.se_filters:
      x = self.se(x, scope, cexp, b.se_filters)
    wp = P.get('%s/%s/kernel' % (scope, 'conv2d_1' if ci else 'conv2d'), (1, 1, cexp, b.output_filters),
               det.conv_kernel_init)
    # the project convolution's gradient buffer is the block output's (engine.bn_res aliases them, or scales by the
    # stochastic-depth mask into a buffer of its own): rounded, not compared
    x = self.bn(self.pw(x, wp, key=scope + ':proj'), '%s/%s' % (scope, bn[bi]), training)
    return self.q(self._residual(x, inputs, b, scope, training), scope + ':out')

  def fused_mbconv(self, inputs, b, scope, training):
    P = self.store
    x = inputs
    cexp = b.input_filters * b.expand_ratio
    k = b.kernel_size
    if b.expand_ratio != 1:
      w = P.get(scope + '/conv2d/kernel', (k, k, b.input_filters, cexp), det.conv_kernel_init)
      x = swish(self.bn(self.conv(x, w, b.stride, scope + ':exp'), scope + '/tpu_batch_normalization', training,
                        scope + ':exp#grad'))
      if b.se_filters:
        x = self.se(x, scope, cexp, b.se_filters)
      wp = P.get(scope + '/conv2d_1/kernel', (1, 1, cexp, b.output_filters), det.conv_kernel_init)
      x = self.bn(self.pw(x, wp, key=scope + ':proj'), scope + '/tpu_batch_normalization_1', training)
      return self.q(self._residual(x, inputs, b, scope, training), scope + ':out')
    w = P.get(scope + '/conv2d/kernel', (k, k, cexp, b.output_filters), det.conv_kernel_init)
    # engine.bn_res with an activated view: d(block output) becomes dz of this BatchNorm IN PLACE in the block output's
    # gradient buffer, so that buffer is compared as dz and the block-output gradient is rounded but not compared
    x = swish(self.bn(self.conv(x, w, b.stride, scope + ':conv'), scope + '/tpu_batch_normalization', training,
                      scope + ':out#grad'))
    return self.q(self._residual(x, inputs, b, scope, training), scope + ':out', None)

  def forward(self, images_nhwc, training=False):
    m = self.mconfig
    name = m.model_name
    P = self.store
    self.new_moving = {}
    stem, blocks = effnetv2_configs.expand_blocks(m)
    x = images_nhwc.permute(0, 3, 1, 2)
    if self.emulate:
      x = x.to(torch.bfloat16).to(torch.float32)
    w = P.get(name + '/stem/conv2d/kernel', (3, 3, 3, stem), det.conv_kernel_init)
    # V2Engine materialises the activated stem output when a dense convolution or a residual reads it (or in training)
    stored = training or blocks[0].has_residual or blocks[0].conv_type == 1
    x = swish(self.bn(self.q(conv2d_same(self.qop(x), self.qop(w), 2), 'stem', None),
                      name + '/stem/tpu_batch_normalization', training, 'stem:out#grad' if stored else 'stem#grad'))
    if stored:
      x = self.q(x, 'stem:out', None)
    ends = {}
    ridx = 0
    for i, b in enumerate(blocks):
      scope = '%s/blocks_%d' % (name, b.index)
      x = (self.mbconv if b.conv_type == 0 else self.fused_mbconv)(x, b, scope, training)
      if i == len(blocks) - 1 or blocks[i + 1].stride > 1:
        ridx += 1
        ends['reduction_%d' % ridx] = x.permute(0, 2, 3, 1)
    ends['features'] = x.permute(0, 2, 3, 1)
    hf = effnetv2_configs.round_filters(m.feature_size or 1280, m)
    wh = P.get(name + '/head/conv2d/kernel', (1, 1, blocks[-1].output_filters, hf), det.conv_kernel_init)
    x = swish(self.bn(self.pw(x, wh, key='head:conv'), name + '/head/tpu_batch_normalization', training,
                      'head:conv#grad'))
    pooled = x.mean(dim=(2, 3))
    ends['pooled_features'] = pooled
    out = pooled
    if self.include_top and m.num_classes:
      wf = P.get(name + '/head/dense/kernel', (hf, m.num_classes), det.zeros)
      bf = P.get(name + '/head/dense/bias', (m.num_classes,), det.zeros)
      if self.emulate:
        # the dense layer is the pointwise kernel on a 1x1 map: the pooled SUMS are cast to bf16, the mean's 1/(H*W) is
        # applied on load and the operand rounded again; bf16 weight copy, fp32 bias, bf16 logits
        hw = float(x.shape[2] * x.shape[3])
        operand = self.qop(self.qop(pooled * hw) * (1.0 / hw))
        out = self.q(operand @ self.qop(wf) + bf, None, None)
      else:
        out = pooled @ wf + bf
    ends['head'] = out
    return ends
