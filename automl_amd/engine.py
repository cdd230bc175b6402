"""Executor of the EfficientDet hot path on one MI355X: launches the gfx950 kernels through the C ABI.

Host side is plain Python (as BASELINE.json's north_star asks): it owns the layer graph, the HBM
buffers (torch tensors are used only as device allocations) and the order of kernel launches.
Design:
  * every conv output is stored ONCE, raw (pre-BatchNorm); BatchNorm + swish + SE gate are applied
    on load by the consumer ("activated view"), BatchNorm statistics come out of the producing
    kernel's epilogue as deterministic per-workgroup partials;
  * backward mirrors it: a data-gradient kernel chains through the consumer-side activation and
    emits the BatchNorm-backward sums; the BatchNorm backward itself is applied on load
    (dy = a*dz + b*y + c) by the producer's wgrad / dgrad kernels;
  * forward records a tape of closures, backward replays it in reverse.
Reference structure followed: efficientdet/tf2/efficientdet_keras.py:787-915 (EfficientDetNet),
efficientdet/backbone/efficientnet_model.py:360-416,710-779, efficientdet/tf2/train_lib.py:493-684.
"""
import contextlib
import ctypes
import re
import math
import os

import numpy as np
import torch

from automl_amd import _lib
from automl_amd import netspec as netspec_lib
from automl_amd import utils
from automl_amd._lib import (ACT_NONE, ACT_SWISH, EDET_BF16, EDET_F32, RS_IDENTITY, RS_POOL, RS_UP2,
                             BwdEpi, GView, TView, call, ptr)


def _pad8(c):
  return (c + 7) // 8 * 8


class Raw(object):
  """A stored NHWC tensor and (lazily) its gradient buffer."""

  def __init__(self, eng, key, n, h, w, c, ld=None, needs_grad=True, dtype=None):
    self.eng, self.key = eng, key
    self.n, self.h, self.w, self.c = n, h, w, c
    self.ld = ld or _pad8(c)
    self.data = eng.buf(key, (n, h, w, self.ld), dtype or eng.tdtype)
    self.needs_grad = needs_grad
    self.grad = None
    self.grad_written = False

  def ensure_grad(self):
    if self.grad is None:
      self.grad = self.eng.buf(self.key + '#grad', (self.n, self.h, self.w, self.ld), self.eng.tdtype)
    return self.grad

  @property
  def rows(self):
    return self.n * self.h * self.w


class BN(object):
  """BatchNorm layer state: parameter slices + per-step derived vectors (fp32 [c])."""

  def __init__(self, eng, name, c):
    self.name, self.c = name, c
    self.gamma = eng.param(name + '/gamma')
    self.beta = eng.param(name + '/beta')
    self.mm = eng.param(name + '/moving_mean')
    self.mv = eng.param(name + '/moving_variance')
    self.dgamma = eng.arena.grad(name + '/gamma')     # per-layer variables: always the main gradient arena
    self.dbeta = eng.arena.grad(name + '/beta')
    v = eng.buf('bn:' + name, (7, c), torch.float32)
    self.scale, self.shift, self.mean, self.rstd, self.a, self.b, self.cc = (v[i] for i in range(7))
    self.count = 0
    self.bwd_ready = False
    self.eval_done = False


class View(object):
  """act(bn(raw)) * gate -- what a consumer kernel sees (edet_tview_t)."""

  def __init__(self, raw, bn=None, act=ACT_NONE, gate=None):
    self.raw, self.bn, self.act, self.gate = raw, bn, act, gate
    self.consumers = 0

  def tview(self):
    r = self.raw
    return TView(ptr(r.data), ptr(self.bn.scale) if self.bn else None,
                 ptr(self.bn.shift) if self.bn else None, ptr(self.gate), self.act,
                 r.n, r.h, r.w, r.c, r.ld)


class ParamArena(object):
  """The model's variables on the device, independent of batch and image size: one flat fp32 master arena of the
  trainable variables (+ gradient, momentum and EMA arenas of the same layout), one of the BatchNorm moving
  statistics, the segment table for per-tensor clipping, and the optimizer's iteration count.  Every Engine of a
  model (one per batch / image shape) works on the SAME arena, so a shape change keeps the optimizer slots, as the
  reference's Keras variables do (train_lib.py:176-199)."""

  def __init__(self, spec, device, values):
    train = [p for p in spec.params if p.trainable]
    state = [p for p in spec.params if not p.trainable]
    self.offsets = {}
    off = 0
    seg = [0]
    flags = []
    for p in train:
      n = int(np.prod(p.shape)) if p.shape else 1
      self.offsets[p.name] = (off, n, p.shape, True)
      off += n
      off_al = (off + 3) // 4 * 4  # keep every tensor 16-byte aligned
      seg.append(off)
      flags.append(1 if netspec_lib.is_l2_regularised(p.name) else 0)
      if off_al != off:
        seg.append(off_al)  # padding segment (zeros, never regularised)
        flags.append(0)
        off = off_al
    self.n_train_elems = off
    self.seg_names = [p.name for p in train]
    soff = 0
    for p in state:
      n = int(np.prod(p.shape))
      self.offsets[p.name] = (soff, n, p.shape, False)
      soff = (soff + n + 3) // 4 * 4
    dev = self.device = torch.device(device)
    self.params_flat = torch.zeros(off, dtype=torch.float32, device=dev)
    self.grads_flat = torch.zeros(off, dtype=torch.float32, device=dev)
    self.velocity = torch.zeros(off, dtype=torch.float32, device=dev)
    self.ema = torch.zeros(off, dtype=torch.float32, device=dev)
    self.state_flat = torch.zeros(max(soff, 4), dtype=torch.float32, device=dev)
    self.seg_offsets = torch.tensor(seg, dtype=torch.int64, device=dev)
    self.seg_flags = torch.tensor(flags, dtype=torch.int32, device=dev)
    self.nseg = len(flags)
    self._l2_flags = list(flags)          # as built: set_frozen starts from these every time
    self.seg_sqnorm = torch.zeros(2 * self.nseg * _lib.OPT_SPLIT, dtype=torch.float32, device=dev)   # norms | L2 partials
    self.seg_factor = torch.ones(self.nseg, dtype=torch.float32, device=dev)
    self.version = 0          # bumped whenever a variable changes: engines re-make their compute copies
    self.step_count = 0       # optimizer iterations applied to this arena
    self.frozen_expr = None
    self.frozen_ranges = []   # [begin, end) element ranges of the frozen variables in the flat arrays (set_frozen)
    self.set_params(values)

  def set_frozen(self, expr):
    """config.var_freeze_expr (tf2/train_lib.py:478-491): the trainable variables whose name -- with the ':0' TensorFlow
    appends -- matches the expression from its start are left out of the L2 term, of the gradient list (per-tensor and
    global clip norms) and of the update.  Here: their segment carries EDET_SEG_FROZEN instead of the L2 flag -- the
    optimizer kernels zero their gradient, count nothing of them in the norms and never touch their value, momentum
    slot or EMA shadow (so an optimizer state restored from an un-frozen run cannot move them either).  The flags are
    rebuilt from the arena's original ones on every call: another expression un-freezes what no longer matches, an
    empty one un-freezes everything.  Returns the frozen names."""
    frozen = []
    if expr:
      pat = re.compile(expr)
      frozen = [n for n in self.seg_names if pat.match(n + ':0')]
    seg_index = {int(o): i for i, o in enumerate(self.seg_offsets.cpu().tolist()[:-1])}
    flags = list(self._l2_flags)
    ranges = []
    for n in frozen:
      off, cnt, _, _ = self.offsets[n]
      flags[seg_index[off]] = _lib.SEG_FROZEN
      if ranges and off - ranges[-1][1] <= 3:      # adjacent up to the alignment padding (zeros): one range
        ranges[-1][1] = off + cnt
      else:
        ranges.append([off, off + cnt])
    self.seg_flags.copy_(torch.tensor(flags, dtype=torch.int32))
    self.frozen_expr = expr or None
    self.frozen_ranges = [(a, b) for a, b in ranges]
    for a, b in self.frozen_ranges:
      self.velocity[a:b].zero_()
    return frozen

  def _slice(self, flat_train, flat_state, name):
    off, n, _, tr = self.offsets[name]
    return (flat_train if tr else flat_state)[off:off + n]

  def param(self, name):
    return self._slice(self.params_flat, self.state_flat, name)

  def grad(self, name):
    off, n, _, tr = self.offsets[name]
    assert tr, name
    return self.grads_flat[off:off + n]

  def set_params(self, values):
    """values: name -> array-like in reference layouts.  Until the first optimizer step the EMA shadow follows the
    variables (TFA MovingAverage seeds the average with the variable's value at its first apply)."""
    for name, v in values.items():
      if name not in self.offsets:
        raise KeyError('unknown variable %s' % name)
      off, n, shape, tr = self.offsets[name]
      t = torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(-1)
      if t.numel() != n:
        raise ValueError('variable %s: expected %d elements, got %d' % (name, n, t.numel()))
      self.param(name).copy_(t)
      if self.step_count == 0 and tr:
        self._slice(self.ema, None, name).copy_(t)
    self.version += 1

  def _export(self, flat_train, flat_state, names):
    out = {}
    for name in (names or self.offsets.keys()):
      shape = self.offsets[name][2]
      out[name] = self._slice(flat_train, flat_state, name).detach().cpu().numpy().reshape(shape)
    return out

  def get_params(self, names=None):
    return self._export(self.params_flat, self.state_flat, names)

  def get_ema_params(self, names=None):
    """The variables as an EMA evaluation would load them: the TFA MovingAverage shadow of every trainable variable
    (train_lib.py:193-197 wraps the optimizer, which averages the variables it updates) and the BatchNorm moving
    statistics as they are (they have no shadow in the Keras train step)."""
    return self._export(self.ema, self.state_flat, names)

  def set_ema_params(self, values):
    """EMA shadows (the MovingAverage optimizer's 'average' slots) of trainable variables, by name."""
    for name, v in values.items():
      if name not in self.offsets:
        raise KeyError('unknown variable %s' % name)
      off, n, shape, tr = self.offsets[name]
      if not tr:
        raise KeyError('%s has no EMA shadow (not a trainable variable)' % name)
      t = torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(-1)
      if t.numel() != n:
        raise ValueError('variable %s: expected %d elements, got %d' % (name, n, t.numel()))
      self._slice(self.ema, None, name).copy_(t)

  def get_optimizer_state(self):
    """Host copy of the optimizer slots: momentum (Adam: first moment), EMA shadows, iteration count; Adam's second moment
    when that optimizer has run."""
    state = {'velocity': self.velocity.cpu().numpy().copy(), 'ema': self.ema.cpu().numpy().copy(),
             'iterations': self.step_count}
    if getattr(self, 'adam_v', None) is not None:
      state['adam_v'] = self.adam_v.cpu().numpy().copy()
    return state

  def set_optimizer_state(self, state):
    self.velocity.copy_(torch.as_tensor(state['velocity']))
    self.ema.copy_(torch.as_tensor(state['ema']))
    if 'adam_v' in state:
      self.second_moment().copy_(torch.as_tensor(state['adam_v']))
    self.step_count = int(state['iterations'])

  def second_moment(self):
    """Adam's v slot, allocated on first use (the SGD configurations never pay for it)."""
    if getattr(self, 'adam_v', None) is None:
      self.adam_v = torch.zeros_like(self.velocity)
    return self.adam_v


class Branch(object):
  """What an independent chain of launches needs of its own: a HIP stream, the BatchNorm partial-sum rows and the
  weight-gradient workspace its kernels scribble on, and -- for a chain that shares variables with another one (the
  class / box towers use the same kernels on every pyramid level, efficientdet_keras.py:336-480) -- a private
  gradient arena that is added to the main one after the join (deterministic, no atomics)."""

  def __init__(self, stream, partials, workspace, grads=None):
    self.stream, self.partials, self.workspace, self.grads = stream, partials, workspace, grads
    self.ws_off = 0       # bytes of the workspace that hold partial sums of deferred reductions (Engine._ws_mark)


class Engine(object):
  """Builds buffers for (config, batch, image size, dtype) and runs forward / backward / update."""
  ADAM_BETA2, ADAM_EPSILON = 0.999, 1e-7      # tf.keras.optimizers.Adam defaults

  def __init__(self, config, batch_size, image_size=None, dtype='bf16', device='cuda:0', seed=0,
               params=None, spec=None, stochastic_depth=True, arena=None):
    if not torch.cuda.is_available():
      raise _lib.EdetError('no HIP device visible: the EfficientDet engine has no CPU path')
    _lib.load()
    self.config = config
    self.spec = spec if spec is not None else netspec_lib.NetSpec(config)
    self.act = getattr(self.spec, 'act_code', ACT_SWISH)     # the model's activation (config.act_type)
    self.bn_momentum = getattr(self.spec, 'bn_momentum', netspec_lib.BN_MOMENTUM)
    self.bn_epsilon = getattr(self.spec, 'bn_epsilon', netspec_lib.BN_EPSILON)
    self.device = torch.device(device)
    torch.cuda.set_device(self.device)
    self.dtype = EDET_BF16 if dtype in ('bf16', EDET_BF16) else EDET_F32
    self.tdtype = torch.bfloat16 if self.dtype == EDET_BF16 else torch.float32
    self.batch = batch_size
    self.image_size = utils.parse_image_size(image_size if image_size is not None else config.image_size)
    self._bufs = {}
    self._zero_list = []
    self._zfree, self._zcur = 0, None
    self.tape = []
    self.training = False
    self._nparts = ctypes.c_int(0)
    self._cast_version = -1
    self._build_params(params, seed, arena)
    cmax = max([p.shape[0] for p in self.spec.params if len(p.shape) == 1] + [64])
    self._cmax = cmax
    # batched weight-gradient reductions (see _ws): EDET_DEFER_REDUCE=0 = one reduction launch per layer, as in round 3
    self.defer_reduce = os.environ.get('EDET_DEFER_REDUCE', '1') != '0'
    self.fuse_wfold = os.environ.get('EDET_FUSE_WFOLD', '1') != '0'
    self._deferring = False
    self._defer_streams = set()
    ws_floats = (64 if self.defer_reduce else 16) * 1024 * 1024      # 256 MiB with deferral (64 MiB: the round-3 scratch)
    self._ws_floats = ws_floats
    self._main = self._branch = Branch(None, torch.empty(_lib.MAX_PARTS * 2 * cmax, dtype=torch.float32, device=self.device),
                                       torch.empty(ws_floats, dtype=torch.float32, device=self.device))
    self._side = None
    # The class / box towers of the SMALL pyramid levels (20x20 and below: ~300 latency-bound launches per step that
    # each use a fraction of the chip) run as one chain on a second HIP stream, forked and joined with events (a
    # parallel branch of the captured hipGraph), under the chain of the 80x80 / 40x40 levels: r02m/n 69.04 -> 67.7 /
    # 68.3 ms per step.  What was measured and NOT kept: one stream per level (five chains, the big kernels of levels
    # 3 and 4 compete: 71.7 -> 75.1 ms, r02g); every weight-gradient kernel deferred to the side stream (nothing on
    # the data-gradient chain waits for them: 68.3 -> 69.7 ms, r02n).
    self.small_level_stream = True
    self._side_pending = False
    self.bns = {}
    self._cast_plan = None
    self.loss_sums = self.zbuf('loss_sums', (4,))
    self.hyper = torch.zeros(4, dtype=torch.float32, device=self.device)   # lr, ema decay, 1/normalizer, -
    self.gnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
    self.pool_argmax = True      # max-pool backward through the recorded winning tap (edet_fuse_bwd_pre)
    self.stochastic_depth = stochastic_depth
    self._cast_items = {}      # weight name -> descriptors of its compute copies (filled by the first pass)
    self._cast_table = None
    self.batched_casts = True    # every compute copy of a step in one edet_cast_batch launch
    # inference forward with bf16 storage: the class / box logits are stored as fp32 (edet_pw_fwd_f32out) -- rounding them to
    # bf16 is 3e-3 of their range and the whole of what separated the path from the 1e-3 of north_star
    # (scripts/precision_sweep.py); the training step keeps bf16 logits (they only feed the loss)
    self.logits_f32 = os.environ.get('EDET_LOGITS_F32', '1') != '0'
    self._f32_island = False
    self.fuse_merge_identity = os.environ.get('EDET_FUSE_MERGE', '1') != '0'   # BiFPN backward: see Engine.fuse
    self.overlap_s2_wgrad = os.environ.get('EDET_S2_OVERLAP', '0') == '1'
    # bucketed gradient all-reduce overlapped with the backward pass (set_overlap_reduce): None = off
    self._overlap_reduce = None
    self._reduce_marks = {}
    self._comm_stream = None
    self._bucket_no = 0
    # head of an MBConv block (expansion -> BatchNorm -> activation -> depthwise) in one kernel where the library has one
    # (mbconv_fused.hip: bf16, <= 32 block-input channels): the expanded tensor is not read back by the depthwise
    # convolution -- and never stored at all in inference.  EDET_MBCONV_FUSED=0: the two-kernel path (lab switch)
    self.fused_mbconv_head = os.environ.get('EDET_MBCONV_FUSED', '1') != '0'
    self.fused_heads = set()     # block scopes whose head ran fused in the last forward pass (inference: no ':exp' tensor)
    self.fused_dw_bwd = True     # one edet_dw_bwd call per layer (bf16: ONE kernel for both gradients, any stride)
    self.fused_pw_bwd = True     # one edet_pw_bwd call per pointwise layer whose input needs a gradient
    # cross-replica BatchNorm (utils.SyncBatchNormalization / TpuBatchNormalization, utils.py:166-241):
    # (all_reduce_fn, world_size) or None.  Set by train_lib when sync_bn=True.
    self.sync_bn = None
    self.bn_bessel = True        # Keras fused BatchNorm: Bessel-corrected batch variance into moving_variance
    self.drop_masks = {}      # block scope -> (mask [n,c] fp32 = floor(p + u_n) / p, survival probability p)
    # optimizer = 'adam' (train_lib.py:183-186): Keras defaults for what the reference does not set
    self.adam = str(getattr(config, 'optimizer', 'sgd')).lower() == 'adam'
    if self.adam:
      self.arena.second_moment()
    self._rng = torch.Generator(device=self.device)
    self._rng.manual_seed(1000003 * seed + 17)

  @property
  def esize(self):
    return 2 if self.dtype == EDET_BF16 else 4

  # ------------------------------------------------------------------ memory
  @property
  def stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  @property
  def partials(self):
    return self._branch.partials

  @property
  def workspace(self):
    return self._branch.workspace

  # ---- deferred weight-gradient reductions (edet_reduce_defer): the backward pass records the ~190 "dW += partial sums"
  # of a step and adds them in a handful of batched launches; until a flush every call gets the workspace BEHIND the
  # partial sums still waiting there
  WS_RESERVE = 64 * 1024 * 1024       # what a single call may need (the round-3 workspace size)

  def _ws(self):
    """(pointer, bytes) of the free part of the current chain's workspace."""
    b = self._branch
    return b.workspace.data_ptr() + b.ws_off, b.workspace.numel() * 4 - b.ws_off

  def _ws_mark(self):
    """After a call that may have left partial sums for a deferred reduction: the next call starts behind them."""
    if not self._deferring:
      return
    b = self._branch
    hi = ctypes.c_void_p()
    call('edet_reduce_deferred_end', self.stream, ctypes.byref(hi))
    if hi.value:
      b.ws_off = (hi.value - b.workspace.data_ptr() + 255) // 256 * 256
      if b.workspace.numel() * 4 - b.ws_off < self.WS_RESERVE:
        call('edet_reduce_flush', self.stream)
        b.ws_off = 0

  def _defer_begin(self):
    if self.defer_reduce and not self._deferring:
      self._deferring = True
      self._defer_streams = set()
    if self._deferring and self.stream not in self._defer_streams:
      call('edet_reduce_defer', self.stream, 1)
      self._defer_streams.add(self.stream)

  def _defer_end(self):
    """Flushes what the current chain recorded and returns its stream to the immediate mode."""
    if self._deferring and self.stream in self._defer_streams:
      call('edet_reduce_defer', self.stream, 0)
      self._defer_streams.discard(self.stream)
      self._branch.ws_off = 0
      if not self._defer_streams:
        self._deferring = False

  # ------------------------------------------------------------------ a second chain on its own stream
  def _side_branch(self):
    if self._side is None:
      names = [n for n in self.seg_names if n.startswith('class_net/') or n.startswith('box_net/')]
      lo = min(self.offsets[n][0] for n in names)
      hi = max(self.offsets[n][0] + self.offsets[n][1] for n in names)
      self._side_range = (lo, hi)       # the slice of the arena that holds the tower variables
      self._side_grads = torch.zeros(self.n_train_elems, dtype=torch.float32, device=self.device)
      self._side = Branch(torch.cuda.Stream(device=self.device),
                          torch.empty(_lib.MAX_PARTS * 2 * self._cmax, dtype=torch.float32, device=self.device),
                          torch.empty(self._ws_floats, dtype=torch.float32, device=self.device),   # as the main one
                          self._side_grads)
    return self._side

  def _fork_join(self, main_job, side_job):
    """main_job() on the current stream, side_job() on the side stream between an event recorded now and an event
    the current stream waits for afterwards (legal inside a stream capture: the side stream joins it)."""
    side = self._side_branch()
    main = torch.cuda.current_stream(self.device)
    rec = _lib.recorder          # a step plan being recorded (automl_amd/plan.py) sees the fork and the join as events
    fork = torch.cuda.Event()
    fork.record(main)
    side.stream.wait_event(fork)
    if rec is not None:
      rec.stream_wait(side.stream.cuda_stream, rec.event_record(main.cuda_stream))
    try:
      self._branch = side
      with torch.cuda.stream(side.stream):
        b = side_job()
        done = torch.cuda.Event()
        done.record(side.stream)
        rec_done = rec.event_record(side.stream.cuda_stream) if rec is not None else None
      self._branch = self._main
      a = main_job()
    finally:
      self._branch = self._main
    main.wait_event(done)
    if rec is not None:
      rec.stream_wait(main.cuda_stream, rec_done)
    return a, b

  def _join_side(self):
    """Adds (and clears) the side chain's gradient arena; called once at the end of the backward pass, after the join."""
    if self._side_pending:
      lo, hi = self._side_range
      call('edet_axpy_clear', self.grads_flat.data_ptr() + 4 * lo, self._side_grads.data_ptr() + 4 * lo, hi - lo, 1,
           self.stream)
      self._side_pending = False

  def buf(self, key, shape, dtype):
    t = self._bufs.get(key)
    if t is None:
      t = torch.empty(shape, dtype=dtype, device=self.device)
      self._bufs[key] = t
    assert tuple(t.shape) == tuple(shape), (key, tuple(t.shape), tuple(shape))
    return t

  ZCHUNK = 4 * 1024 * 1024      # floats per zero-arena chunk (16 MiB)

  def zbuf(self, key, shape):
    """fp32 buffer that is zeroed at the start of every step (atomic accumulation targets: SE pooled sums and gate
    gradients, fusion-weight gradients, loss sums).  Carved out of a few large chunks so that the ~60 buffers of a
    step cost one fill launch per chunk instead of one each."""
    t = self._bufs.get(key)
    if t is None:
      n = int(np.prod(shape))
      padded = (n + 63) // 64 * 64                   # every buffer starts 256-byte aligned
      if padded > self.ZCHUNK:
        chunk = torch.zeros(padded, dtype=torch.float32, device=self.device)
        self._zero_list.append(chunk)
        t = chunk[:n].view(shape)
      else:
        if self._zcur is None or self._zfree + padded > self.ZCHUNK:
          self._zero_list.append(torch.zeros(self.ZCHUNK, dtype=torch.float32, device=self.device))
          self._zfree = 0
          self._zcur = self._zero_list[-1]
        t = self._zcur[self._zfree:self._zfree + n].view(shape)
        self._zfree += padded
      self._bufs[key] = t
    return t

  def _build_params(self, params, seed, arena):
    if arena is None:
      values = params if params is not None else netspec_lib.init_params(self.spec, seed)
      if any(p.name not in values for p in self.spec.params):
        # a partial set (e.g. a checkpoint restored with skip_mismatch): the other variables keep their initial values
        values = {**netspec_lib.init_params(self.spec, seed), **values}
      arena = ParamArena(self.spec, self.device, values)
    self.arena = arena
    for k in ('offsets', 'n_train_elems', 'seg_names', 'params_flat', 'grads_flat', 'velocity', 'ema', 'state_flat',
              'seg_offsets', 'seg_flags', 'nseg', 'seg_sqnorm', 'seg_factor'):
      setattr(self, k, getattr(arena, k))

  @property
  def step_count(self):
    return self.arena.step_count

  @property
  def _cast_dirty(self):
    return self._cast_version != self.arena.version

  def param(self, name):
    return self.arena.param(name)

  def grad(self, name):
    """Gradient slice of a variable in the arena the current chain accumulates into (Branch.grads)."""
    if self._branch.grads is not None:
      off, n, _, tr = self.offsets[name]
      assert tr, name
      return self._branch.grads[off:off + n]
    return self.arena.grad(name)

  def set_params(self, values):
    """values: name -> array-like in reference layouts."""
    self.arena.set_params(values)

  def get_params(self, names=None):
    return self.arena.get_params(names)

  def get_grads(self):
    return {name: self.grad(name).detach().cpu().numpy().reshape(self.offsets[name][2])
            for name in self.seg_names}

  # ------------------------------------------------------------------ compute copies of 1x1 kernels
  def _pw_copies(self, name, cin, cout):
    """(Wt [cout][ldk], ldk, W [cin][ldn], ldn) compute-dtype copies of an HWIO 1x1 kernel."""
    ldk, ldn = _pad8(cin), _pad8(cout)
    sfx = ':f32' if self._f32_island else ''      # fp32 copies of a layer that runs in fp32 inside a bf16 engine
    wt = self.buf('wt:' + name + sfx, (cout, ldk), self.tdtype)
    w = self.buf('w:' + name + sfx, (cin, ldn), self.tdtype)
    if name + sfx not in self._cast_done:
      src = ptr(self.param(name))
      call('edet_cast_matrix', src, ptr(wt), cin, cout, ldk, 1, self.dtype, self.stream)
      call('edet_cast_matrix', src, ptr(w), cin, cout, ldn, 0, self.dtype, self.stream)
      self._cast_done.add(name + sfx)
      if not sfx:        # (the batched re-cast of a step makes the engine's own compute type only)
        self._cast_items[name] = [(src, ptr(wt), cin, cout, ldk, 1), (src, ptr(w), cin, cout, ldn, 0)]
    return wt, ldk, w, ldn

  @contextlib.contextmanager
  def _fp32_island(self):
    """Layers built inside run through the fp32 kernels of the library although the engine stores bf16 (their inputs
    must be fp32 tensors: Engine._to_f32).  Inference only -- the box-predict layer, see _head_level."""
    saved = (self.dtype, self.tdtype, self._f32_island)
    self.dtype, self.tdtype, self._f32_island = EDET_F32, torch.float32, True
    try:
      yield
    finally:
      self.dtype, self.tdtype, self._f32_island = saved

  def _to_f32(self, key, v):
    """fp32 copy of a stored tensor, same view (BatchNorm / activation are applied on load by the fp32 kernels)."""
    r = v.raw
    out = Raw(self, key, r.n, r.h, r.w, r.c, r.ld, needs_grad=False, dtype=torch.float32)
    call('edet_cast_to_f32', ptr(r.data), ptr(out.data), r.data.numel(), EDET_BF16 if r.data.dtype == torch.bfloat16 else EDET_F32,
         self.stream)
    return View(out, v.bn, v.act, v.gate)

  def _cast_all(self):
    """Every compute copy recorded by an earlier pass, re-made in one launch (edet_cast_batch)."""
    names = list(self._cast_items)
    if self._cast_table is None or self._cast_table[0] != names:
      if torch.cuda.is_current_stream_capturing():
        return                 # no host-to-device copy inside a capture: this pass casts layer by layer
      import struct
      flat = [it for n in names for it in self._cast_items[n]]
      raw = b''.join(struct.pack('<QQiiii', s or 0, d, r, c, ld, t) for (s, d, r, c, ld, t) in flat)
      dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
      biggest = max((t and c or r) * ld for (_, _, r, c, ld, t) in flat)
      self._cast_table = (names, dev, len(flat), max(1, min(64, (biggest + 255) // 256)))
    _, dev, count, blocks = self._cast_table
    call('edet_cast_batch', ptr(dev), count, blocks, self.dtype, self.stream)
    self._cast_done.update(names)

  # ------------------------------------------------------------------ BatchNorm plumbing
  def get_bn(self, name, c):
    bn = self.bns.get(name)
    if bn is None:
      bn = BN(self, name, c)
      self.bns[name] = bn
    return bn

  def _sync_partials(self, nparts, c):
    """Cross-replica sum of the [nparts][2][c] partial rows -> one row at partials[0:2c] (every replica
    normalises with the statistics of the GLOBAL batch: mean of the shard means / mean squares over equal
    shards, utils.py:176-195,215-241).  Returns the local sums."""
    reduce_fn, _ = self.sync_bn
    local = self.partials[:nparts * 2 * c].view(nparts, 2 * c).sum(0)
    total = local.clone()
    reduce_fn(total)
    self.partials[:2 * c].copy_(total)
    return local

  def _bn_forward(self, bn, count, nparts):
    if self.training:
      if self.sync_bn is not None:
        self._sync_partials(nparts, bn.c)
        nparts, count = 1, count * self.sync_bn[1]
      bn.count = count
      call('edet_bn_finalize', ptr(self.partials), nparts, bn.c, float(count), ptr(bn.gamma), ptr(bn.beta),
           self.bn_epsilon, self.bn_momentum if self.update_moving else -1.0, 1 if (self.bn_bessel and self.sync_bn is None) else 0,
           ptr(bn.mm), ptr(bn.mv), ptr(bn.scale), ptr(bn.shift), ptr(bn.mean), ptr(bn.rstd), self.stream)
      bn.bwd_ready = False
      bn.eval_done = False
      self.arena.version += 1          # moving statistics (and soon the weights) change
    elif not bn.eval_done:
      call('edet_bn_eval', bn.c, ptr(bn.gamma), ptr(bn.beta), self.bn_epsilon, ptr(bn.mm), ptr(bn.mv),
           ptr(bn.scale), ptr(bn.shift), self.stream)
      bn.eval_done = True

  def _bn_bwd_finalize(self, bn, nparts):
    dgamma, dbeta = ptr(bn.dgamma), ptr(bn.dbeta)
    if self.sync_bn is not None:
      # gamma / beta gradients from the LOCAL sums (the gradient all-reduce adds the replicas up), the
      # on-load coefficients (a, b, cc) from the GLOBAL sums over the global count
      call('edet_bn_bwd_finalize', ptr(self.partials), nparts, bn.c, float(bn.count), ptr(bn.gamma),
           ptr(bn.mean), ptr(bn.rstd), dgamma, dbeta, None, ptr(bn.a), ptr(bn.b), ptr(bn.cc), self.stream)
      self._sync_partials(nparts, bn.c)
      nparts, dgamma, dbeta = 1, None, None
    call('edet_bn_bwd_finalize', ptr(self.partials), nparts, bn.c, float(bn.count), ptr(bn.gamma),
         ptr(bn.mean), ptr(bn.rstd), dgamma, dbeta, None, ptr(bn.a), ptr(bn.b), ptr(bn.cc),
         self.stream)
    bn.bwd_ready = True

  def _ensure_bn_bwd(self, v):
    """Make the on-load BatchNorm-backward coefficients of view v available (multi-consumer case)."""
    if v.bn is None or v.bn.bwd_ready:
      return
    r = v.raw
    call('edet_bn_bwd_reduce', ptr(r.grad), ptr(r.data), r.rows, r.c, r.ld, ptr(v.bn.mean), ptr(v.bn.rstd),
         ptr(self.partials), ctypes.byref(self._nparts), self.dtype, self.stream)
    self._bn_bwd_finalize(v.bn, self._nparts.value)

  def _gview(self, v):
    """dy of the conv that produced v.raw, as an on-load gradient view."""
    self._ensure_bn_bwd(v)
    r = v.raw
    if v.bn is not None:
      return GView(ptr(r.grad), ptr(r.data), ptr(v.bn.a), ptr(v.bn.b), ptr(v.bn.cc), r.n, r.h, r.w, r.c, r.ld)
    return GView(ptr(r.grad), None, None, None, None, r.n, r.h, r.w, r.c, r.ld)

  def _epi(self, vin, dgate=None):
    """Epilogue descriptor for writing d(vin) into vin.raw.grad; returns (epi, fused_stats)."""
    r = vin.raw
    g = r.ensure_grad()
    beta = 1 if r.grad_written else 0
    fused = (vin.bn is not None and vin.consumers == 1 and vin.gate is None and beta == 0)
    epi = BwdEpi(ptr(g), beta,
                 ptr(vin.bn.mean) if fused else None, ptr(vin.bn.rstd) if fused else None,
                 ptr(self.partials) if fused else None, ptr(dgate))
    return epi, fused

  # ------------------------------------------------------------------ layers
  def pw(self, key, vin, wname, cout, bias=None, bn=None, act=ACT_NONE, ld=None, f32out=False, bias_grad=False):
    """1x1 conv (+bias) [-> BN -> act as a view].  f32out (inference, bf16 storage): the output is stored as fp32 --
    the class / box logits, whose bf16 rounding alone is 3e-3 of their range (Engine.logits_f32)."""
    r = vin.raw
    cin = r.c
    wt, ldk, w, ldn = self._pw_copies(wname, cin, cout)
    if f32out and not self.training and self.dtype == EDET_BF16 and bn is None:
      out = Raw(self, key + ':f32', r.n, r.h, r.w, cout, ld, needs_grad=False, dtype=torch.float32)
      call('edet_pw_fwd_f32out', ctypes.byref(vin.tview()), ptr(wt), ldk, ptr(self.param(bias)) if bias else None,
           ptr(out.data), cout, out.ld, self.stream, nbytes=r.rows * (cin * 2 + cout * 4),
           tag='%dx%dx%d->%d f32' % (r.h, r.w, cin, cout))
      vin.consumers += 1
      return View(out, None, act)
    out = Raw(self, key, r.n, r.h, r.w, cout, ld)
    bnl = self.get_bn(bn, cout) if bn else None
    stats = ptr(self.partials) if (bnl and self.training) else None
    call('edet_pw_fwd', ctypes.byref(vin.tview()), ptr(wt), ldk, ptr(self.param(bias)) if bias else None,
         ptr(out.data), cout, out.ld, stats, ctypes.byref(self._nparts), self.dtype, self.stream,
         nbytes=r.rows * (cin + cout) * self.esize, tag='%dx%dx%d->%d' % (r.h, r.w, cin, cout))
    if bnl:
      self._bn_forward(bnl, out.rows, self._nparts.value)
    vout = View(out, bnl, act)
    vin.consumers += 1
    if self.training:
      self.tape.append(lambda: self._pw_bwd(vin, vout, wname, w, ldn, bias is None))
      if bias_grad and bias and bnl is None:
        self.tape.append(lambda: self._bias_grad(vout, bias))      # (replayed BEFORE _pw_bwd: the gradient is complete by then)
    return vout

  def _bias_grad(self, vout, bias):
    """d(bias) += column sums of the output gradient, for a biased convolution with NO BatchNorm behind it that is not a
    predict layer (those get theirs from the loss kernels; behind a BatchNorm the gradient is analytically zero): the
    resample convolutions under apply_bn_for_resampling=False.  The BatchNorm-backward pair with mean 0 / rstd 1 / gamma 1:
    its dbeta is exactly that sum."""
    r = vout.raw
    assert r.grad is not None and r.grad_written, 'bias gradient of %s before its output gradient is complete' % bias
    # main chain only: the constant vector is filled once, on the stream that is current at its creation, and nothing
    # orders a later reader on ANOTHER stream against that fill (the resample convolutions are BiFPN layers)
    assert self._branch is self._main, 'bias gradient of %s on the side chain' % bias
    fresh = ('ones:c:%d' % r.c) not in self._bufs
    ones = self.buf('ones:c:%d' % r.c, (r.c,), torch.float32)
    if fresh:
      ones.fill_(1.0)          # (once, at creation: not a launch of every replayed step)
    zeros = self.zbuf('zeros:c:%d' % r.c, (r.c,))
    scr = self.buf('biasgrad:scr:%d' % r.c, (3, r.c), torch.float32)
    call('edet_bn_bwd_reduce', ptr(r.grad), ptr(r.data), r.rows, r.c, r.ld, ptr(zeros), ptr(ones), ptr(self.partials),
         ctypes.byref(self._nparts), self.dtype, self.stream)
    call('edet_bn_bwd_finalize', ptr(self.partials), self._nparts.value, r.c, float(r.rows), ptr(ones), ptr(zeros),
         ptr(ones), None, ptr(self.grad(bias)), None, ptr(scr[0]), ptr(scr[1]), ptr(scr[2]), self.stream)

  def _pw_bwd(self, vin, vout, wname, w, ldn, no_bias=False):
    g = self._gview(vout)
    nb = vin.raw.rows * (vin.raw.c + vout.raw.c) * self.esize
    tag = '%dx%dx%d->%d' % (vin.raw.h, vin.raw.w, vin.raw.c, vout.raw.c)
    if vin.raw.needs_grad and self.fused_pw_bwd:
      # both gradients in one call: one pass over (dz, y, x) where the layer fits the fused kernel
      dgate = vin.dgate if vin.gate is not None else None
      epi, fused = self._epi(vin, dgate)
      if no_bias:
        epi.flags = _lib.EPI_Y_IS_CONV_OF_INPUT      # y = view(x) W exactly: the library need not read it
      call('edet_pw_bwd', ctypes.byref(g), ptr(w), ldn, ctypes.byref(vin.tview()), ctypes.byref(epi),
           ctypes.byref(self._nparts), ptr(self.grad(wname)), *self._ws(),
           self.dtype, self.stream, nbytes=2 * nb, tag=tag)
      self._ws_mark()
      vin.raw.grad_written = True
      if fused:
        self._bn_bwd_finalize(vin.bn, self._nparts.value)
      return
    call('edet_pw_bwd_weight', ctypes.byref(vin.tview()), ctypes.byref(g), ptr(self.grad(wname)),
         *self._ws(), self.dtype, self.stream, nbytes=nb, tag=tag)
    self._ws_mark()
    if vin.raw.needs_grad:
      dgate = vin.dgate if vin.gate is not None else None
      epi, fused = self._epi(vin, dgate)
      call('edet_pw_bwd_data', ctypes.byref(g), ptr(w), ldn, ctypes.byref(vin.tview()), ctypes.byref(epi),
           ctypes.byref(self._nparts), self.dtype, self.stream, nbytes=nb, tag=tag)
      vin.raw.grad_written = True
      if fused:
        self._bn_bwd_finalize(vin.bn, self._nparts.value)

  def conv(self, key, vin, wname, k, stride, cout, bn=None, act=ACT_NONE):
    """Dense k x k convolution, TF 'SAME', no bias [-> BN -> act as a view] (Fused-MBConv,
    efficientnetv2/effnetv2_model.py:338-346,362-371).  Forward only: the V2 classifier's training
    is outside the hot path, so the tape entry refuses to run."""
    r = vin.raw
    cin = r.c
    if cin % 8 != 0 or r.ld != cin:
      raise ValueError('dense convolution needs an input channel count divisible by 8, got %d' % cin)
    kk = k * k * cin
    wt = self.buf('wtc:' + wname, (cout, kk), self.tdtype)
    if wname not in self._cast_done:
      call('edet_cast_matrix', ptr(self.param(wname)), ptr(wt), kk, cout, kk, 1, self.dtype, self.stream)
      self._cast_done.add(wname)
    oh, _, _ = utils.same_padding(r.h, k, stride)
    ow, _, _ = utils.same_padding(r.w, k, stride)
    out = Raw(self, key, r.n, oh, ow, cout)
    bnl = self.get_bn(bn, cout) if bn else None
    stats = ptr(self.partials) if (bnl and self.training) else None
    call('edet_conv_fwd', ctypes.byref(vin.tview()), ptr(wt), kk, k, stride, ptr(out.data), cout, out.ld,
         stats, ctypes.byref(self._nparts), self.dtype, self.stream,
         nbytes=(r.rows * cin + out.rows * cout) * self.esize,
         tag='%dx%dx%d->%d k%ds%d' % (r.h, r.w, cin, cout, k, stride))
    if bnl:
      self._bn_forward(bnl, out.rows, self._nparts.value)
    vout = View(out, bnl, act)
    vin.consumers += 1
    if self.training:
      self.tape.append(lambda: self._conv_bwd(vin, vout, wname, k, stride, cin, cout))
    return vout

  def _conv_bwd(self, vin, vout, wname, k, stride, cin, cout):
    """Both gradients of a dense convolution (edet_conv_bwd_weight / edet_conv_bwd_data)."""
    g = self._gview(vout)
    nb = (vin.raw.rows * cin + vout.raw.rows * cout) * self.esize
    tag = '%dx%dx%d->%d k%ds%d' % (vin.raw.h, vin.raw.w, cin, cout, k, stride)
    call('edet_conv_bwd_weight', ctypes.byref(vin.tview()), ctypes.byref(g), k, stride, ptr(self.grad(wname)),
         *self._ws(), self.dtype, self.stream, nbytes=nb, tag=tag)
    self._ws_mark()
    if vin.raw.needs_grad:
      # compute copy for the data gradient: HWIO -> [cin][k][k][cout] (reduction index (tap, co) contiguous)
      wperm = self.buf('wtd:' + wname, (cin, k * k * cout), self.tdtype)
      wperm.copy_(self.param(wname).view(k, k, cin, cout).permute(2, 0, 1, 3).reshape(cin, k * k * cout))
      epi, fused = self._epi(vin)
      call('edet_conv_bwd_data', ctypes.byref(g), ptr(wperm), k * k * cout, k, stride, ctypes.byref(vin.tview()),
           ctypes.byref(epi), ctypes.byref(self._nparts), self.dtype, self.stream, nbytes=nb, tag=tag)
      vin.raw.grad_written = True
      if fused:
        self._bn_bwd_finalize(vin.bn, self._nparts.value)

  def dw(self, key, vin, wname, k, stride, bn=None, act=ACT_NONE):
    r = vin.raw
    oh, _, _ = utils.same_padding(r.h, k, stride)
    ow, _, _ = utils.same_padding(r.w, k, stride)
    out = Raw(self, key, r.n, oh, ow, r.c)
    bnl = self.get_bn(bn, r.c) if bn else None
    stats = ptr(self.partials) if (bnl and self.training) else None
    wp = ptr(self.param(wname))
    call('edet_dw_fwd', ctypes.byref(vin.tview()), wp, k, stride, ptr(out.data), out.ld, stats,
         ctypes.byref(self._nparts), self.dtype, self.stream, nbytes=(r.rows + out.rows) * r.c * self.esize,
         tag='%dx%dx%d k%ds%d' % (r.h, r.w, r.c, k, stride))
    if bnl:
      self._bn_forward(bnl, out.rows, self._nparts.value)
    vout = View(out, bnl, act)
    vin.consumers += 1
    if self.training:
      self.tape.append(lambda: self._dw_bwd(vin, vout, wname, k, stride))
    return vout

  def _dw_bwd(self, vin, vout, wname, k, stride):
    g = self._gview(vout)
    nb = (vin.raw.rows + vout.raw.rows) * vin.raw.c * self.esize
    tag = '%dx%dx%d k%ds%d' % (vin.raw.h, vin.raw.w, vin.raw.c, k, stride)
    if vin.raw.needs_grad and self.fused_dw_bwd:
      epi, fused = self._epi(vin)
      call('edet_dw_bwd', ctypes.byref(g), ptr(self.param(wname)), k, stride, ctypes.byref(vin.tview()),
           ctypes.byref(epi), ctypes.byref(self._nparts), ptr(self.grad(wname)), *self._ws(), self.dtype, self.stream, nbytes=2 * nb, tag=tag)
      self._ws_mark()
      vin.raw.grad_written = True
      if fused:
        self._bn_bwd_finalize(vin.bn, self._nparts.value)
      return
    gptr = ptr(self.grad(wname))

    def wgrad():
      call('edet_dw_bwd_weight', ctypes.byref(vin.tview()), ctypes.byref(g), k, stride, gptr,
           *self._ws(), self.dtype, self.stream, nbytes=nb, tag=tag)
      self._ws_mark()

    def dgrad():
      epi, fused = self._epi(vin)
      call('edet_dw_bwd_data', ctypes.byref(g), ptr(self.param(wname)), k, stride, ctypes.byref(vin.tview()),
           ctypes.byref(epi), ctypes.byref(self._nparts), self.dtype, self.stream, nbytes=nb, tag=tag)
      vin.raw.grad_written = True
      if fused:
        self._bn_bwd_finalize(vin.bn, self._nparts.value)

    if vin.raw.needs_grad and self.overlap_s2_wgrad and self.training and self.sync_bn is None and self._branch is self._main:
      # Stride-2 layer: the weight-gradient kernel (nothing on the chain waits for it) on the side stream NEXT TO the
      # data-gradient kernel -- both march over the same (dz, y, x), the second reader finds them in the L2 / MALL
      # (Engine.overlap_s2_wgrad; the side chain records its reduction on its own stream and flushes before the join)
      def side_job():
        self._defer_begin()
        try:
          wgrad()
        finally:
          self._defer_end()
      self._fork_join(dgrad, side_job)
      return
    wgrad()
    if vin.raw.needs_grad:
      dgrad()

  def se(self, key, v, scope, se_filters):
    """Squeeze-and-excitation: returns the gated view of v (efficientnet_model.py:183-195)."""
    r = v.raw
    n, c = r.n, r.c
    inv_hw = 1.0 / (r.h * r.w)
    pooled = self.buf(key + ':pool', (n, c), torch.float32)
    hidden = self.buf(key + ':hid', (n, se_filters), torch.float32)
    gate = self.buf(key + ':gate', (n, c), torch.float32)
    w1, b1 = scope + '/se/conv2d/kernel', scope + '/se/conv2d/bias'
    w2, b2 = scope + '/se/conv2d_1/kernel', scope + '/se/conv2d_1/bias'
    # pooling (chunk sums in the BatchNorm partial-row scratch, free between two layers) + both 1x1 layers; no atomics,
    # batch-independent summation order
    call('edet_se_squeeze_excite', ctypes.byref(v.tview()), ptr(self.partials), self.partials.numel() * 4,
         se_filters, inv_hw, ptr(self.param(w1)), ptr(self.param(b1)), ptr(self.param(w2)), ptr(self.param(b2)),
         ptr(pooled), ptr(hidden), ptr(gate), v.act, self.dtype, self.stream, nbytes=r.rows * c * self.esize)
    vg = View(r, v.bn, v.act, gate)
    vg.dgate = self.zbuf(key + ':dgate', (n, c)) if self.training else None
    v.consumers += 1
    if self.training:
      dpool = self.buf(key + ':dpool', (n, c), torch.float32)
      scratch = self.buf(key + ':scr', (n * (c + (2 + (c + 127) // 128) * se_filters) +
                                        8 * (2 * c * se_filters + c + se_filters),), torch.float32)

      def bwd():
        call('edet_se_fc_bwd', ptr(pooled), ptr(hidden), ptr(gate), ptr(vg.dgate), n, c, se_filters, inv_hw,
             ptr(self.param(w1)), ptr(self.param(w2)), ptr(self.grad(w1)), ptr(self.grad(b1)),
             ptr(self.grad(w2)), ptr(self.grad(b2)), ptr(dpool), ptr(scratch), v.act, self.stream)
        call('edet_se_gate_bwd', ctypes.byref(vg.tview()), ptr(r.grad), ptr(dpool), ptr(v.bn.mean),
             ptr(v.bn.rstd), ptr(self.partials), ctypes.byref(self._nparts), self.dtype, self.stream,
             nbytes=2 * r.rows * c * self.esize)
        self._bn_bwd_finalize(v.bn, self._nparts.value)

      self.tape.append(bwd)
    return vg

  def refresh_drop_masks(self):
    """New stochastic-depth draws: mask[n, :] = floor(p + u_n) / p, u_n ~ U[0,1) per image
    (utils.drop_connect, utils.py:329-344).  Device-side torch ops, outside any captured graph."""
    for mask, p in self.drop_masks.values():
      u = torch.rand(mask.shape[0], 1, device=self.device, generator=self._rng)
      mask.copy_(((u + p).floor() / p).expand_as(mask))

  def bn_res(self, key, vy, residual, survival_prob=None):
    """Materialise a block output: bn(y) [* stochastic-depth scale] (+ residual)."""
    r = vy.raw
    if residual is not None and (residual.bn is not None or residual.act != ACT_NONE or residual.gate is not None):
      raise ValueError('bn_res: the residual operand must be a stored (plain) tensor')
    out = Raw(self, key, r.n, r.h, r.w, r.c)
    mask = None
    if self.training and self.stochastic_depth and survival_prob and residual is not None:
      if key not in self.drop_masks:
        self.drop_masks[key] = (self.buf(key + ':dc', (r.n, r.c), torch.float32), float(survival_prob))
        if not torch.cuda.is_current_stream_capturing():
          saved = self.drop_masks
          self.drop_masks = {key: saved[key]}
          self.refresh_drop_masks()
          self.drop_masks = saved
      mask = self.drop_masks[key][0]
      vy.consumers += 1
      vy = View(r, vy.bn, vy.act, mask)
    call('edet_bn_res', ctypes.byref(vy.tview()), ptr(residual.raw.data) if residual else None, ptr(out.data),
         out.ld, self.dtype, self.stream, nbytes=(3 if residual else 2) * r.rows * r.c * self.esize)
    vout = View(out)
    vy.consumers += 1
    if residual is not None:
      residual.consumers += 1
    if self.training:
      act_view = vy.act != ACT_NONE

      def bwd():
        assert out.grad_written, key
        if act_view:
          # out = act(bn(y)) [* mask] (+ residual): the residual takes d(out) as it is, then d(out) becomes
          # dz = d(out) * mask * act'(z) in place (edet_se_gate_bwd with gate = mask or ones, dpool = 0), which
          # also yields the BatchNorm-backward sums of y's BatchNorm
          if residual is not None and residual.raw.needs_grad:
            rr = residual.raw
            call('edet_add', ptr(rr.ensure_grad()), ptr(out.grad), rr.rows, rr.c, rr.ld,
                 1 if rr.grad_written else 0, self.dtype, self.stream)
            rr.grad_written = True
          ones = mask if mask is not None else self.buf('ones:%d:%d' % (r.n, r.c), (r.n, r.c), torch.float32)
          if mask is None:
            ones.fill_(1.0)
          zeros = self.zbuf('zeros:%d:%d' % (r.n, r.c), (r.n, r.c))
          gv = TView(ptr(r.data), ptr(vy.bn.scale), ptr(vy.bn.shift), ptr(ones), vy.act, r.n, r.h, r.w, r.c, r.ld)
          call('edet_se_gate_bwd', ctypes.byref(gv), ptr(out.grad), ptr(zeros), ptr(vy.bn.mean), ptr(vy.bn.rstd),
               ptr(self.partials), ctypes.byref(self._nparts), self.dtype, self.stream,
               nbytes=2 * r.rows * r.c * self.esize)
          self._bn_bwd_finalize(vy.bn, self._nparts.value)
          r.grad = out.grad
          r.grad_written = True
          return
        if mask is not None:
          # d(bn output) = d(block output) * mask[n]: the same kernel, applied to the gradient
          gbuf = self.buf(key + ':dcg', (r.n, r.h, r.w, r.ld), self.tdtype)
          gv = TView(ptr(out.grad), None, None, ptr(mask), ACT_NONE, r.n, r.h, r.w, r.c, out.ld)
          call('edet_bn_res', ctypes.byref(gv), None, ptr(gbuf), r.ld, self.dtype, self.stream,
               nbytes=2 * r.rows * r.c * self.esize)
          r.grad = gbuf
        else:
          r.grad = out.grad         # d(bn output) aliases d(block output)
        r.grad_written = True
        if residual is not None and residual.raw.needs_grad:
          rr = residual.raw
          call('edet_add', ptr(rr.ensure_grad()), ptr(out.grad), rr.rows, rr.c, rr.ld,
               1 if rr.grad_written else 0, self.dtype, self.stream)
          rr.grad_written = True
      self.tape.append(bwd)
    return vout

  def fuse(self, key, inputs, modes, wnames, oh, ow, act=ACT_SWISH):
    """BiFPN node fusion: act(sum_i wn_i * resample_i(input_i))."""
    c = inputs[0].raw.c
    n = inputs[0].raw.n
    nin = len(inputs)
    out = Raw(self, key, n, oh, ow, c)
    wm = self.spec.fpn.weight_method
    wc = c if (wnames and wm.startswith('channel_')) else 1       # per-channel weight vectors (WSM shape [c])
    wn = self.buf(key + ':wn', (max(4, 3 * wc),), torch.float32)
    method = (2 if wm in ('attn', 'channel_attn') else 0) if wnames else 1
    wp = [ptr(self.param(w)) for w in wnames] + [None] * (3 - len(wnames)) if wnames else [None] * 3
    # scalar fusion variables: normalised inside the fusion kernel, their gradient inside the ordered finish of dwn (no
    # edet_fuse_weights / edet_fuse_weights_bwd launches: 50 of them per D0 step; EDET_FUSE_WFOLD=0: separate launches)
    wfold = self.fuse_wfold and wc == 1
    wraw = (ctypes.c_void_p * 3)(*wp) if wfold else None
    if not wfold:
      call('edet_fuse_weights', wp[0], wp[1], wp[2], nin, method, ptr(wn), wc, self.stream)
    tv = [v.tview() for v in inputs]
    tvp = [ctypes.byref(t) for t in tv] + [None] * (3 - nin)
    marr = (ctypes.c_int * 3)(*(list(modes) + [0] * (3 - nin)))
    fbytes = (sum(v.raw.rows for v in inputs) + out.rows) * c * self.esize
    call('edet_fuse_fwd', tvp[0], tvp[1], tvp[2], marr, nin, ptr(wn), wc, act, ptr(out.data), oh, ow, out.ld,
         wraw, method, self.dtype, self.stream, nbytes=fbytes)
    vout = View(out)
    for v in inputs:
      v.consumers += 1
    if self.training:
      ds = self.buf(key + ':ds', (n, oh, ow, out.ld), self.tdtype)
      dwn = self.zbuf(key + ':dwn', (max(4, 3 * wc),))
      npool = sum(1 for m in modes if m == RS_POOL)
      amax = self.buf(key + ':amax', (npool, n, oh, ow, c), torch.uint8) if npool and self.pool_argmax else None

      def bwd():
        assert out.grad_written, key
        tv2 = [v.tview() for v in inputs]
        tvp2 = [ctypes.byref(t) for t in tv2] + [None] * (3 - nin)
        # identity inputs that need a gradient: written by the fusion kernel itself (no edet_fuse_bwd_input launch, no
        # second read of ds); ds is stored only when a resampled input still has to read it
        merged = [self.fuse_merge_identity and modes[i] == RS_IDENTITY and v.raw.needs_grad for i, v in enumerate(inputs)]
        gin = (ctypes.c_void_p * 3)()
        gbeta = (ctypes.c_int * 3)()
        for i, v in enumerate(inputs):
          if merged[i]:
            gin[i] = ptr(v.raw.ensure_grad())
            gbeta[i] = 1 if v.raw.grad_written else 0
            v.raw.grad_written = True
        write_ds = 1 if any(v.raw.needs_grad and not merged[i] for i, v in enumerate(inputs)) else 0
        wgrad_folded = wfold and bool(wnames) and method != 1
        dwraw = (ctypes.c_void_p * 3)(*([ptr(self.grad(w)) for w in wnames] + [None] * (3 - len(wnames)))) if wgrad_folded else None
        call('edet_fuse_bwd_pre', tvp2[0], tvp2[1], tvp2[2], marr, nin, ptr(wn), wc, act, ptr(out.grad), oh, ow,
             out.ld, ptr(ds), ptr(dwn), ptr(amax), gin, gbeta, write_ds, *self._ws(),
             wraw if wgrad_folded else None, method, dwraw, self.dtype, self.stream,
             nbytes=fbytes + out.rows * c * self.esize)
        plane = 0
        for i, v in enumerate(inputs):
          am = None
          if modes[i] == RS_POOL and amax is not None:
            am = amax[plane].data_ptr()
            plane += 1
          if not v.raw.needs_grad or merged[i]:
            continue
          g = v.raw.ensure_grad()
          call('edet_fuse_bwd_input', ctypes.byref(tv2[i]), modes[i], ptr(wn), wc, i, ptr(ds), oh, ow, out.ld, am,
               ptr(g), 1 if v.raw.grad_written else 0, self.dtype, self.stream,
               nbytes=(out.rows + v.raw.rows) * c * self.esize,
               tag='%dx%dx%d %s' % (v.raw.h, v.raw.w, c, ('id', 'up2', 'pool')[modes[i]]))
          v.raw.grad_written = True
        if wnames and not wgrad_folded:
          gp = [ptr(self.grad(w)) for w in wnames] + [None] * (3 - len(wnames))
          call('edet_fuse_weights_bwd', wp[0], wp[1], wp[2], nin, method, ptr(dwn), gp[0], gp[1], gp[2],
               wc, self.stream)

      self.tape.append(bwd)
    return vout

  # ------------------------------------------------------------------ network
  def _begin(self, training, update_moving=True):
    self.training = training
    self.update_moving = update_moving
    self.tape = []
    self.fused_heads = set()
    # compute copies of the kernels and the inference BatchNorm vectors are rebuilt only when a variable
    # changed since they were made (every training step; in inference only after set_params)
    if self._cast_dirty or not hasattr(self, '_cast_done'):
      self._cast_done = set()
      if self._cast_items and self.batched_casts:
        self._cast_all()
      for bn in self.bns.values():
        bn.eval_done = False
      self._cast_version = self.arena.version
    for bn in self.bns.values():
      bn.bwd_ready = False
    for t in self._zero_list:      # atomic accumulation targets (SE pooled sums, ...) start every pass at zero
      call('edet_zero', ptr(t), t.numel() * 4, self.stream)
    if training:
      call('edet_zero', ptr(self.grads_flat), self.grads_flat.numel() * 4, self.stream)

  def forward(self, images, training=False, update_moving=True):
    """images: device tensor [B,H,W,3] in the engine dtype. Returns (cls_views, box_views)."""
    c = self.config
    spec = self.spec
    assert tuple(images.shape) == (self.batch, self.image_size[0], self.image_size[1], 3), images.shape
    assert images.dtype == self.tdtype and images.is_contiguous()
    self._begin(training, update_moving)
    if training and self.drop_masks and not torch.cuda.is_current_stream_capturing():
      self.refresh_drop_masks()
    self.images = images
    n, h, w = self.batch, self.image_size[0], self.image_size[1]
    bb = c.backbone_name
    # ---- stem
    oh, _, _ = utils.same_padding(h, 3, 2)
    ow, _, _ = utils.same_padding(w, 3, 2)
    y0 = Raw(self, 'stem', n, oh, ow, spec.stem_filters)
    bn0 = self.get_bn(bb + '/stem/tpu_batch_normalization', spec.stem_filters)
    wstem = bb + '/stem/conv2d/kernel'
    call('edet_stem_fwd', ptr(images), n, h, w, ptr(self.param(wstem)), ptr(y0.data), spec.stem_filters, y0.ld,
         ptr(self.partials) if training else None, ctypes.byref(self._nparts), self.dtype, self.stream,
         nbytes=(n * h * w * 3 + y0.rows * y0.c) * self.esize)
    self._bn_forward(bn0, y0.rows, self._nparts.value)
    x = View(y0, bn0, self.act)
    if training:
      v0 = x

      def stem_bwd():
        g = self._gview(v0)
        call('edet_stem_bwd_weight', ptr(images), n, h, w, ctypes.byref(g), ptr(self.grad(wstem)), *self._ws(), self.dtype,
             self.stream, nbytes=(n * h * w * 3 + y0.rows * y0.c) * self.esize)
        self._ws_mark()
      self.tape.append(stem_bwd)
    # ---- MBConv blocks
    reds = []
    self._reduce_marks = {}
    bucket_blocks = self._bucket_blocks() if (training and self._overlap_reduce is not None) else ()
    for b in spec.blocks:
      if b.index in bucket_blocks:
        # everything the tape holds from here on belongs to variables at or behind this block's first one
        self._reduce_marks[len(self.tape)] = bucket_blocks[b.index]
      x = self._mbconv(x, b, '%s/blocks_%d' % (bb, b.index))
      if b.index in spec.reductions:
        reds.append(x)
    all_feats = [None] + reds
    feats = list(all_feats[c.min_level:c.max_level + 1])
    wf = c.fpn_num_filters
    if training and self._overlap_reduce is not None:
      self._reduce_marks[0] = 0                                     # stem + first blocks: the last bucket
      self._reduce_marks[len(self.tape)] = self._first_non_backbone_elem()   # BiFPN + heads: the first one
    # ---- extra levels P6.. (efficientdet_keras.py:823-836,900-901)
    for level in range(6, c.max_level + 1):
      f = feats[-1]
      th, tw = (f.raw.h + 1) // 2, (f.raw.w + 1) // 2
      s = 'resample_p%d' % level
      if f.raw.c != wf and getattr(c, 'conv_after_downsample', False):
        # ResampleFeatureMap with conv_after_downsample (efficientdet_keras.py:316-324): the 1x1 convolution (+ BN) runs on
        # the POOLED map -- a quarter of the rows -- instead of before the pool
        pooled = self.fuse(s + ':pool', [f], [RS_POOL], [], th, tw, act=ACT_NONE)
        feats.append(self.pw(s, pooled, s + '/conv2d/kernel', wf, bias=s + '/conv2d/bias', bn=(s + '/bn') if c.apply_bn_for_resampling else None, bias_grad=True))
        continue
      if f.raw.c != wf:
        f = self.pw(s, f, s + '/conv2d/kernel', wf, bias=s + '/conv2d/bias', bn=(s + '/bn') if c.apply_bn_for_resampling else None, bias_grad=True)
      feats.append(self.fuse(s + ':pool', [f], [RS_POOL], [], th, tw, act=ACT_NONE))
    # ---- BiFPN
    for rep in range(c.fpn_cell_repeats):
      feats = self._fpn_cell(feats, 'fpn_cells/cell_%d' % rep)
    self.fpn_feats = feats
    # ---- heads
    na = spec.num_anchors
    # side chain = the levels of at most 20x20 pixels (r02q: also moving the 40x40 level there 68.3 -> 68.5 ms, only the
    # 10x10 and 5x5 levels 69.0 ms)
    nbig = sum(1 for f in feats if f.raw.h * f.raw.w > 400)
    if self.small_level_stream and self.sync_bn is None and 0 < nbig < len(feats):
      cls, box = self._heads_two_chains(feats, nbig, c.num_classes * na, 4 * na)
    else:
      cls = self._head(feats, 'class_net', 'class', c.num_classes * na)
      box = self._head(feats, 'box_net', 'box', 4 * na)
    self.cls_views, self.box_views = cls, box
    if self.batched_casts and self._cast_items and not torch.cuda.is_current_stream_capturing() and \
        (self._cast_table is None or self._cast_table[0] != list(self._cast_items)):
      self._cast_all()         # builds the descriptor table now (outside any capture); the casts it repeats are idempotent
    return cls, box

  def _mbconv(self, xin, b, scope):
    cexp = b.input_filters * b.expand_ratio
    bn_names = ['tpu_batch_normalization', 'tpu_batch_normalization_1', 'tpu_batch_normalization_2']
    conv_names = ['conv2d', 'conv2d_1']
    bi = ci = 0
    x = xin
    if b.expand_ratio != 1 and self._mbconv_head_fusable(x, cexp, b.kernel_size, b.stride):
      x = self.mbconv_head(scope, x, '%s/%s/kernel' % (scope, conv_names[ci]), cexp, '%s/%s' % (scope, bn_names[bi]),
                           scope + '/depthwise_conv2d/depthwise_kernel', b.kernel_size, b.stride,
                           '%s/%s' % (scope, bn_names[bi + 1]))
      ci += 1
      bi += 2
    else:
      if b.expand_ratio != 1:
        x = self.pw(scope + ':exp', x, '%s/%s/kernel' % (scope, conv_names[ci]), cexp,
                    bn='%s/%s' % (scope, bn_names[bi]), act=self.act)
        ci += 1
        bi += 1
      x = self.dw(scope + ':dw', x, scope + '/depthwise_conv2d/depthwise_kernel', b.kernel_size, b.stride,
                  bn='%s/%s' % (scope, bn_names[bi]), act=self.act)
      bi += 1
    if b.se_filters:
      x = self.se(scope + ':se', x, scope, b.se_filters)
    y = self.pw(scope + ':proj', x, '%s/%s/kernel' % (scope, conv_names[ci]), b.output_filters,
                bn='%s/%s' % (scope, bn_names[bi]), act=ACT_NONE)
    sps = getattr(self.spec, 'survival_probs', None)
    return self.bn_res(scope + ':out', y, xin if b.has_residual else None,
                       survival_prob=sps[b.index] if sps else None)

  def _mbconv_head_fusable(self, vin, cexp, k, stride):
    if not self.fused_mbconv_head or self.dtype != EDET_BF16 or self.act == ACT_NONE:
      return False
    return _lib.load().edet_mbconv_fused_supported(ctypes.byref(vin.tview()), cexp, k, stride, self.dtype) == 1

  def mbconv_head(self, scope, vin, wexp, cexp, bn_exp, wdw, k, stride, bn_dw):
    """x = act(bn0(expand_conv(x))); x = act(bn1(depthwise_conv(x))) (efficientnet_model.py:378-392) through
    edet_mbconv_expand_dw_fwd.  Training: the expansion's batch statistics first (edet_mbconv_expand_stats, the block
    input only), the raw expanded tensor stored for the backward pass, whose tape entries are those of the two-kernel
    path (Engine.pw, Engine.dw).  Inference: the expanded tensor is never stored."""
    r = vin.raw
    cin = r.c
    self.fused_heads.add(scope)
    wt, ldk, w, ldn = self._pw_copies(wexp, cin, cexp)
    bne, bnd = self.get_bn(bn_exp, cexp), self.get_bn(bn_dw, cexp)
    oh, _, _ = utils.same_padding(r.h, k, stride)
    ow, _, _ = utils.same_padding(r.w, k, stride)
    tv = vin.tview()
    tag = '%dx%dx%d->%d k%ds%d' % (r.h, r.w, cin, cexp, k, stride)
    e_raw = None
    if self.training:
      e_raw = Raw(self, scope + ':exp', r.n, r.h, r.w, cexp)
      call('edet_mbconv_expand_stats', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(self.partials),
           ctypes.byref(self._nparts), self.dtype, self.stream, nbytes=r.rows * cin * self.esize, tag=tag)
    self._bn_forward(bne, r.rows, self._nparts.value)
    out = Raw(self, scope + ':dw', r.n, oh, ow, cexp)
    stats = ptr(self.partials) if self.training else None
    call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(bne.scale), ptr(bne.shift), self.act,
         ptr(e_raw.data) if e_raw is not None else None, e_raw.ld if e_raw is not None else 0,
         ptr(self.param(wdw)), k, stride, ptr(out.data), out.ld, stats, ctypes.byref(self._nparts), self.dtype, self.stream,
         nbytes=(r.rows * cin + (r.rows * cexp if e_raw is not None else 0) + out.rows * cexp) * self.esize, tag=tag)
    self._bn_forward(bnd, out.rows, self._nparts.value)
    vout = View(out, bnd, self.act)
    vin.consumers += 1
    if self.training:
      ve = View(e_raw, bne, self.act)
      ve.consumers = 1
      self.tape.append(lambda: self._pw_bwd(vin, ve, wexp, w, ldn, True))
      self.tape.append(lambda: self._dw_bwd(ve, vout, wdw, k, stride))
    return vout

  def _fpn_cell(self, feats, cell_scope):
    c = self.config
    wf = c.fpn_num_filters
    fpn = self.spec.fpn
    feats = list(feats)
    num_in = len(feats)
    for n, node in enumerate(fpn.nodes):
      scope = '%s/fnode%d' % (cell_scope, n)
      lvl = node['feat_level'] - c.min_level
      th, tw = feats[lvl].raw.h, feats[lvl].raw.w
      ins, modes = [], []
      for i, off in enumerate(node['inputs_offsets']):
        f = feats[off]
        if f.raw.c != wf:
          if getattr(c, 'conv_after_downsample', False) and f.raw.h > th and f.raw.w > tw:
            # (no BiFPN of fpn_configs.py feeds a node a wider AND larger map; the extra levels P6.. are handled above)
            raise ValueError('conv_after_downsample inside a BiFPN node is not built')
          rs = '%s/resample_%d_%d_%d' % (scope, i, off, len(feats))
          f = self.pw(rs, f, rs + '/conv2d/kernel', wf, bias=rs + '/conv2d/bias', bn=(rs + '/bn') if c.apply_bn_for_resampling else None, bias_grad=True)
        fh, fw = f.raw.h, f.raw.w
        if fh > th and fw > tw:
          if (fh - 1) // th + 1 != 2 or (fw - 1) // tw + 1 != 2:
            raise ValueError('only 2x down-sampling between pyramid levels is supported')
          modes.append(RS_POOL)
        elif fh <= th and fw <= tw:
          modes.append(RS_IDENTITY if (fh == th and fw == tw) else RS_UP2)
        else:
          raise ValueError('Incompatible Resampling : feat shape {}x{} target_shape: {}x{}'.format(
              fh, fw, th, tw))
        ins.append(f)
      wnames = []
      if fpn.weight_method in ('fastattn', 'attn', 'channel_fastattn', 'channel_attn'):
        wnames = [scope + '/WSM' + ('' if i == 0 else '_%d' % i) for i in range(len(ins))]
      x = self.fuse(scope + ':fuse', ins, modes, wnames, th, tw, act=self.act)
      oc = '%s/op_after_combine%d' % (scope, len(feats))
      d = self.dw(oc + ':dw', x, oc + '/conv/depthwise_kernel', 3, 1)
      y = self.pw(oc + ':pw', d, oc + '/conv/pointwise_kernel', wf, bias=oc + '/conv/bias', bn=oc + '/bn')
      feats.append(y)
    out = []
    for level in range(c.min_level, c.max_level + 1):
      for i, node in enumerate(reversed(fpn.nodes)):
        if node['feat_level'] == level:
          out.append(feats[-1 - i])
          break
    assert len(out) == num_in
    return out

  def _head_level(self, feat, level, net, prefix, out_ch):
    """One tower (class or box net) on one pyramid level (efficientdet_keras.py:336-480, 483-641)."""
    c = self.config
    wf = c.fpn_num_filters
    x = feat
    sp = getattr(c, 'survival_prob', None)
    for i in range(c.box_class_repeats):
      s = '%s/%s-%d' % (net, prefix, i)
      key = '%s:l%d' % (s, level)
      d = self.dw(key + ':dw', x, s + '/depthwise_kernel', 3, 1)
      y = self.pw(key + ':pw', d, s + '/pointwise_kernel', wf, bias=s + '/bias',
                  bn='%s/%s-%d-bn-%d' % (net, prefix, i, level), act=self.act)
      if sp:
        # config.survival_prob (efficientdet_keras.py:434-436, 612-614): from the second tower layer on
        # image = drop_connect(act(bn(conv(image)))) + image -- a residual connection in inference too, a per-image
        # floor(p + u) / p scale on the branch in training (utils.py:329-344; own draws per layer, level and tower).  The
        # residual operand is a stored tensor, so the first layer's activated output is materialised as well.
        x = self.bn_res(key + ':out', y, x if i > 0 else None, survival_prob=sp if i > 0 else None)
      else:
        x = y
    s = '%s/%s-predict' % (net, prefix)
    key = '%s:l%d' % (s, level)
    if net == 'box_net' and self.logits_f32 and not self.training and self.dtype == EDET_BF16:
      # Inference, bf16 storage: the BOX-predict layer (depthwise 3x3 + 64 -> 36 pointwise) runs in fp32.  Error budget
      # of the box logits against the fp32 oracle (scripts/precision_sweep.py, d0 640x640): bf16 matrix-core operands of
      # this one layer 1.1e-3 of the range, its depthwise output stored as bf16 6.1e-4, everything else together 3e-4
      # -- the class logits are at 1e-4 with bf16 operands and need no such treatment.
      with self._fp32_island():
        d = self.dw(key + ':dw:f32', self._to_f32(key + ':x:f32', x), s + '/depthwise_kernel', 3, 1)
        return self.pw(key + ':pw:f32', d, s + '/pointwise_kernel', out_ch, bias=s + '/bias')
    d = self.dw(key + ':dw', x, s + '/depthwise_kernel', 3, 1)
    return self.pw(key + ':pw', d, s + '/pointwise_kernel', out_ch, bias=s + '/bias', f32out=self.logits_f32)

  def _head(self, feats, net, prefix, out_ch):
    return [self._head_level(feat, self.config.min_level + li, net, prefix, out_ch) for li, feat in enumerate(feats)]

  def _heads_two_chains(self, feats, nbig, cls_ch, box_ch):
    """Both towers as two chains: the big levels (the first nbig) on the current stream, the small ones on the side
    stream (Engine.small_level_stream); the backward pass replays each chain's tape the same way, the side chain's
    gradients of the shared tower kernels going to its own arena, which _join_side adds after the join."""
    c = self.config
    for net, prefix, out_ch in (('class_net', 'class', cls_ch), ('box_net', 'box', box_ch)):
      # compute copies of the shared tower kernels are made BEFORE the fork (the chain that would otherwise make them
      # first runs concurrently with the one that uses them)
      for i in range(c.box_class_repeats):
        self._pw_copies('%s/%s-%d/pointwise_kernel' % (net, prefix, i), c.fpn_num_filters, c.fpn_num_filters)
      self._pw_copies('%s/%s-predict/pointwise_kernel' % (net, prefix), c.fpn_num_filters, out_ch)
    if self.logits_f32 and not self.training and self.dtype == EDET_BF16:
      # ... and the fp32 copies of the box-predict kernel (_head_level's fp32 island): the chain that casts them first would
      # otherwise do so on ITS stream while the other chain reads the same buffer with no event in between
      with self._fp32_island():
        self._pw_copies('box_net/box-predict/pointwise_kernel', c.fpn_num_filters, box_ch)
    main_tape = self.tape
    tapes = {}
    # which chain runs which (level, tower): the big levels on the main chain, the small ones on the side chain.  Both
    # towers of a level stay on ONE chain: they accumulate into the same feature gradient, and only stream order
    # orders the beta = 0 writer before the beta = 1 one (r05 lab: splitting a level lost 0.5 ms anyway)
    towers = (('class_net', 'class', cls_ch), ('box_net', 'box', box_ch))
    work = {'main': [], 'side': []}
    for li in range(len(feats)):
      for tw in towers:
        on_main = li < nbig
        work['main' if on_main else 'side'].append((li, tw))

    def chain(which):
      def job():
        self.tape = []
        out = {}
        for li, (net, prefix, out_ch) in work[which]:
          out[li, net] = self._head_level(feats[li], c.min_level + li, net, prefix, out_ch)
        tapes[which] = self.tape
        return out
      return job
    try:
      done, side_done = self._fork_join(chain('main'), chain('side'))
    finally:
      self.tape = main_tape
    done.update(side_done)
    outs = [(done[li, 'class_net'], done[li, 'box_net']) for li in range(len(feats))]
    if self.training:
      def bwd():
        def replay(which):
          def job():
            self._defer_begin()        # (the side chain records on its own stream and flushes before the join)
            try:
              for fn in reversed(tapes[which]):
                fn()
            finally:
              if which == 'side':
                self._defer_end()
          return job
        self._fork_join(replay('main'), replay('side'))
        self._side_pending = True
      self.tape.append(bwd)
    return [o[0] for o in outs], [o[1] for o in outs]

  # ------------------------------------------------------------------ outputs
  def outputs(self):
    """(list of [B,h,w,A*classes], list of [B,h,w,A*4]) strided torch views of the logits buffers."""
    cls = [v.raw.data[..., :v.raw.c] for v in self.cls_views]
    box = [v.raw.data[..., :v.raw.c] for v in self.box_views]
    return cls, box

  # ------------------------------------------------------------------ loss + backward + update
  def loss_backward(self, labels):
    """Detection loss forward+backward (train_lib.py:493-604) then the tape in reverse.

    labels: dict with device tensors cls_targets_L int32 [B,h,w,A], box_targets_L fp32 [B,h,w,4A],
    mean_num_positives fp32 [B] (or [B,1]).
    """
    c = self.config
    assert self.training
    norm_dev = None
    if labels.get('normalizer') == 'device':
      # 1/normalizer lives in self.hyper[2] (set_normalizer): nothing of the step depends on a host value,
      # so the launches below can be captured once and replayed for every batch
      normalizer, norm_dev = 1.0, ptr(self.hyper[2:])
    elif 'normalizer' in labels:   # host float supplied by the caller: no device sync
      normalizer = float(labels['normalizer'])
    else:
      normalizer = float(labels['mean_num_positives'].sum().item()) + 1.0
    na = self.spec.num_anchors
    for li, (cv, bv) in enumerate(zip(self.cls_views, self.box_views)):
      level = c.min_level + li
      ct = labels['cls_targets_%d' % level]
      bt = labels['box_targets_%d' % level]
      assert ct.dtype == torch.int32 and ct.is_contiguous() and bt.dtype == torch.float32 and bt.is_contiguous()
      r = cv.raw
      ls = float(getattr(c, 'label_smoothing', 0.0) or 0.0)
      if ls:      # FocalLoss(label_smoothing), train_lib.py:400-402
        call('edet_focal_loss_smooth', ptr(r.data), r.ld, ptr(ct), r.rows, na, c.num_classes, c.alpha, c.gamma, ls,
             1.0 / normalizer, norm_dev, ptr(r.ensure_grad()), ptr(self.grad('class_net/class-predict/bias')),
             ptr(self.loss_sums), *self._ws(), self.dtype, self.stream,
             nbytes=2 * r.rows * r.c * self.esize)
      else:
        call('edet_focal_loss', ptr(r.data), r.ld, ptr(ct), r.rows, na, c.num_classes, c.alpha, c.gamma,
             1.0 / normalizer, norm_dev, ptr(r.ensure_grad()), ptr(self.grad('class_net/class-predict/bias')),
             ptr(self.loss_sums), *self._ws(), self.dtype, self.stream,
             nbytes=2 * r.rows * r.c * self.esize)
      r.grad_written = True
      rb = bv.raw
      call('edet_box_loss', ptr(rb.data), rb.ld, ptr(bt), rb.rows, 4 * na, c.delta, 1.0 / (normalizer * 4.0),
           float(c.box_loss_weight), norm_dev, ptr(rb.ensure_grad()), ptr(self.grad('box_net/box-predict/bias')),
           ptr(self.loss_sums), *self._ws(), self.dtype, self.stream)
      rb.grad_written = True
    self.backward()

  def backward(self):
    self._defer_begin()          # the weight-gradient sums of this pass: recorded, added in a few batched launches
    marks = self._reduce_marks if self._overlap_reduce is not None else {}
    self._bucket_hi = self.n_train_elems
    self._bucket_no = 0
    try:
      for i in range(len(self.tape) - 1, -1, -1):
        self.tape[i]()
        if i in marks:
          self._reduce_bucket(marks[i])
    finally:
      self._defer_end()
    self.tape = []
    self._join_side()
    if marks:
      self._finish_buckets()

  # ---- gradient all-reduce overlapped with the backward pass (north_star; legal only without the global-norm clip) ----
  def set_overlap_reduce(self, reduce_fn, buckets=6):
    """reduce_fn(flat_slice) = in-place SUM all-reduce.  The reference clips the LOCAL gradient by its GLOBAL norm before
    apply_gradients reduces it (train_lib.py:675-683), which needs every gradient of the step: with clip_gradients_norm > 0
    nothing may be reduced before the backward pass has ended.  With clip_gradients_norm = 0 (hparams_config.py:220 allows
    it) the local gradient of a variable is final as soon as its layer's backward has run, so the arena is reduced in
    `buckets` contiguous ranges in the order the backward pass completes them -- BiFPN + heads first, then the backbone
    stages last to first (the arena is laid out in forward order) -- on a communication stream, under the backward
    kernels of the earlier layers.  The L2 gradient (train_lib.py:486-491) is added per range just before its reduce."""
    clip = abs(self.config.clip_gradients_norm) if self.config.clip_gradients_norm else 0.0
    if reduce_fn is not None and clip > 0:
      raise ValueError('overlapping the gradient all-reduce with the backward pass needs clip_gradients_norm=0: the '
                       'reference clips the local gradient by its global norm BEFORE the reduce (train_lib.py:675-683)')
    self._overlap_reduce = reduce_fn
    self._overlap_buckets = int(buckets)
    if reduce_fn is not None and self._comm_stream is None:
      self._comm_stream = torch.cuda.Stream(device=self.device)
      self._seg_host = [int(o) for o in self.arena.seg_offsets.cpu().tolist()]
      self._bucket_gn = torch.zeros(64, dtype=torch.float32, device=self.device)

  def _first_non_backbone_elem(self):
    bb = self.config.backbone_name + '/'
    return min(self.offsets[n][0] for n in self.seg_names if not n.startswith(bb))

  def _bucket_blocks(self):
    """{block index: first arena element of the block} for the blocks that open a backbone bucket: walking the blocks
    from the last one, a bucket is closed once it holds its share of the backbone's elements."""
    bb = self.config.backbone_name
    first = {}
    for b in self.spec.blocks:
      pre = '%s/blocks_%d/' % (bb, b.index)
      first[b.index] = min(self.offsets[n][0] for n in self.seg_names if n.startswith(pre))
    end = self._first_non_backbone_elem()
    share = end / max(self._overlap_buckets - 1, 1)
    out, hi = {}, end
    for idx in sorted(first, reverse=True):
      if hi - first[idx] >= share and first[idx] > 0:
        out[idx] = first[idx]
        hi = first[idx]
    return out

  def _reduce_bucket(self, lo):
    """The arena range [lo, previous lo) is final: flush its deferred weight-gradient sums, add the L2 gradient, and hand it
    to the communication stream."""
    hi = self._bucket_hi
    if lo >= hi:
      return
    self._defer_end()                 # the partial sums recorded so far -> the arena
    self._join_side()                 # (first bucket: the side chain's share of the tower gradients)
    seg = self._seg_host
    s0, s1 = seg.index(lo), seg.index(hi)
    c = self.config
    no = self._bucket_no
    sq = self.buf('ovl:sq%d' % no, (2 * (s1 - s0) * _lib.OPT_SPLIT,), torch.float32)
    st = self.stream
    call('edet_opt_l2_norms', ptr(self.grads_flat), ptr(self.params_flat), self.seg_offsets.data_ptr() + 8 * s0,
         self.seg_flags.data_ptr() + 4 * s0, s1 - s0, float(c.weight_decay), ptr(sq), st)
    call('edet_opt_clip_factors', ptr(sq), s1 - s0, 0.0, self.seg_factor.data_ptr() + 4 * s0,
         self._bucket_gn.data_ptr() + 4 * no, ptr(self.loss_sums[2:]), st)
    main = torch.cuda.current_stream(self.device)
    ready = torch.cuda.Event()
    ready.record(main)
    self._comm_stream.wait_event(ready)
    with torch.cuda.stream(self._comm_stream):
      self._overlap_reduce(self.grads_flat[lo:hi])
    self._bucket_hi = lo
    self._bucket_no = no + 1
    self._defer_begin()

  def _finish_buckets(self):
    assert self._bucket_hi == 0, self._bucket_hi
    done = torch.cuda.Event()
    done.record(self._comm_stream)
    torch.cuda.current_stream(self.device).wait_event(done)
    gn = self._bucket_gn[:self._bucket_no]
    self.gnorm.copy_(torch.sqrt((gn * gn).sum()).reshape(1))      # the local gradient's norm (reported only)

  def set_hyper(self, lr, ema_decay=None):
    """Per-step scalars -> device (hyper[0] = learning rate, hyper[1] = EMA decay).  Stream-ordered H2D
    copy from pageable memory (staged synchronously by the runtime, so the host values may change at once);
    kept OUTSIDE the captured step."""
    if self.adam:
      # tf.keras Adam's bias-corrected rate of THIS step (t = iterations + 1), ResourceApplyAdam's alpha
      t = self.arena.step_count + 1
      b1 = float(self.config.momentum)
      lr = lr * math.sqrt(1.0 - self.ADAM_BETA2 ** t) / (1.0 - b1 ** t)
    self.hyper[:2].copy_(torch.tensor([lr, ema_decay or 0.0], dtype=torch.float32), non_blocking=True)

  def set_normalizer(self, mean_num_positives):
    """hyper[2] = 1 / (sum(mean_num_positives) + 1) computed on the device (train_lib.py:517), no host sync."""
    m = mean_num_positives.reshape(-1)
    if m.dtype != torch.float32 or not m.is_contiguous():
      m = m.float().contiguous()
    call('edet_loss_normalizer', ptr(m), m.numel(), self.hyper.data_ptr() + 8, self.stream)

  def optimizer_local(self, scale_for_reduce):
    """L2 (train_lib.py:486-491) + per-tensor and global-norm clip factors of the LOCAL gradient (:675-682);
    scale_for_reduce applies the factors in place (the data-parallel path all-reduces the clipped gradient)."""
    c = self.config
    st = self.stream
    # (frozen variables -- ParamArena.set_frozen -- are handled by the kernels through their segment flag)
    call('edet_opt_l2_norms', ptr(self.grads_flat), ptr(self.params_flat), ptr(self.seg_offsets),
         ptr(self.seg_flags), self.nseg, float(c.weight_decay), ptr(self.seg_sqnorm), st)
    clip = abs(c.clip_gradients_norm) if c.clip_gradients_norm else 0.0
    call('edet_opt_clip_factors', ptr(self.seg_sqnorm), self.nseg, float(clip), ptr(self.seg_factor),
         ptr(self.gnorm), ptr(self.loss_sums[2:]), st)
    if scale_for_reduce:
      call('edet_opt_scale', ptr(self.grads_flat), ptr(self.seg_offsets), ptr(self.seg_factor), self.nseg, st)

  def optimizer_apply(self, use_ema, already_scaled):
    """SGD momentum (or Adam) + EMA (train_lib.py:176-199) with lr / decay from self.hyper (set_hyper)."""
    if self.adam:
      call('edet_opt_adam_ema', ptr(self.params_flat), ptr(self.grads_flat), ptr(self.velocity), ptr(self.arena.second_moment()),
           ptr(self.ema) if use_ema else None, ptr(self.seg_offsets), None if already_scaled else ptr(self.seg_factor),
           ptr(self.seg_flags), self.nseg, ptr(self.hyper), float(self.config.momentum), self.ADAM_BETA2, self.ADAM_EPSILON,
           self.stream)
      self.arena.version += 1
      self.arena.step_count += 1
      return
    call('edet_opt_sgd_ema', ptr(self.params_flat), ptr(self.grads_flat), ptr(self.velocity),
         ptr(self.ema) if use_ema else None, ptr(self.seg_offsets),
         None if already_scaled else ptr(self.seg_factor), ptr(self.seg_flags), self.nseg, ptr(self.hyper),
         float(self.config.momentum), self.stream)
    self.arena.version += 1
    self.arena.step_count += 1

  def optimizer_step(self, lr, ema_decay=None, all_reduce=None):
    """L2 + clip (local, before the reduce) + [all-reduce SUM] + SGD momentum + EMA."""
    self.set_hyper(lr, ema_decay)
    if self._overlap_reduce is not None:       # L2 + reduce already done range by range inside backward()
      self.optimizer_apply(ema_decay is not None, True)
      return
    self.optimizer_local(all_reduce is not None)
    if all_reduce is not None:
      all_reduce(self.grads_flat)
    self.optimizer_apply(ema_decay is not None, all_reduce is not None)

  def loss_values(self):
    s = self.loss_sums.detach().cpu().numpy()
    c = self.config
    det = float(s[0] + c.box_loss_weight * s[1])
    return {'cls_loss': float(s[0]), 'box_loss': float(s[1]), 'det_loss': det, 'reg_l2_loss': float(s[2]),
            'loss': det + float(s[2]), 'gradient_norm': float(self.gnorm.item())}
