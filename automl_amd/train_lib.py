"""Training step of the detection path on MI355X (mirror of efficientdet/tf2/train_lib.py).

``EfficientDetNetTrain.train_step((images, labels))`` follows train_lib.py:606-684: forward with
training BatchNorm, focal + Huber detection loss (:493-604), L2 on kernels (:486-491), per-tensor
clip_by_norm then clip_by_global_norm (:675-682, on the LOCAL gradient), gradient all-reduce SUM
across data-parallel replicas (implicit in optimizer.apply_gradients, :683), SGD momentum and the
TFA MovingAverage EMA (:176-199).  LR schedules restate :37-173.
"""
import math
import os

import torch

from automl_amd import efficientdet_net


def update_learning_rate_schedule_parameters(params):
  """train_lib.py:37-49 (params: dict-like with batch_size and steps_per_epoch)."""
  params['adjusted_learning_rate'] = params['learning_rate'] * params['batch_size'] / 64
  spe = params['steps_per_epoch']
  params['lr_warmup_step'] = int(params['lr_warmup_epoch'] * spe)
  params['first_lr_drop_step'] = int(params['first_lr_drop_epoch'] * spe)
  params['second_lr_drop_step'] = int(params['second_lr_drop_epoch'] * spe)
  params['total_steps'] = int(params['num_epochs'] * spe)


def _warmup(step, lr_warmup_init, lr_warmup_step, adjusted_lr):
  return lr_warmup_init + (float(step) / lr_warmup_step * (adjusted_lr - lr_warmup_init))


class StepwiseLrSchedule(object):
  def __init__(self, adjusted_lr, lr_warmup_init, lr_warmup_step, first_lr_drop_step, second_lr_drop_step):
    self.adjusted_lr, self.lr_warmup_init, self.lr_warmup_step = adjusted_lr, lr_warmup_init, lr_warmup_step
    self.first_lr_drop_step, self.second_lr_drop_step = first_lr_drop_step, second_lr_drop_step

  def __call__(self, step):
    lr = _warmup(step, self.lr_warmup_init, self.lr_warmup_step, self.adjusted_lr) \
        if step < self.lr_warmup_step else self.adjusted_lr
    for mult, start in ((1.0, self.lr_warmup_step), (0.1, self.first_lr_drop_step),
                        (0.01, self.second_lr_drop_step)):
      if step >= start:
        lr = self.adjusted_lr * mult
    return lr


class CosineLrSchedule(object):
  def __init__(self, adjusted_lr, lr_warmup_init, lr_warmup_step, total_steps):
    self.adjusted_lr, self.lr_warmup_init, self.lr_warmup_step = adjusted_lr, lr_warmup_init, lr_warmup_step
    self.decay_steps = float(total_steps - lr_warmup_step)

  def __call__(self, step):
    if step < self.lr_warmup_step:
      return _warmup(step, self.lr_warmup_init, self.lr_warmup_step, self.adjusted_lr)
    return 0.5 * self.adjusted_lr * (1 + math.cos(math.pi * float(step) / self.decay_steps))


class PolynomialLrSchedule(object):
  def __init__(self, adjusted_lr, lr_warmup_init, lr_warmup_step, power, total_steps):
    self.adjusted_lr, self.lr_warmup_init, self.lr_warmup_step = adjusted_lr, lr_warmup_init, lr_warmup_step
    self.power, self.total_steps = power, total_steps

  def __call__(self, step):
    if step < self.lr_warmup_step:
      return _warmup(step, self.lr_warmup_init, self.lr_warmup_step, self.adjusted_lr)
    return self.adjusted_lr * (1 - float(step) / self.total_steps)**self.power


def learning_rate_schedule(params):
  update_learning_rate_schedule_parameters(params)
  m = params['lr_decay_method']
  if m == 'stepwise':
    return StepwiseLrSchedule(params['adjusted_learning_rate'], params['lr_warmup_init'],
                              params['lr_warmup_step'], params['first_lr_drop_step'],
                              params['second_lr_drop_step'])
  if m == 'cosine':
    return CosineLrSchedule(params['adjusted_learning_rate'], params['lr_warmup_init'],
                            params['lr_warmup_step'], params['total_steps'])
  if m == 'polynomial':
    return PolynomialLrSchedule(params['adjusted_learning_rate'], params['lr_warmup_init'],
                                params['lr_warmup_step'], params['poly_lr_power'], params['total_steps'])
  raise ValueError('unknown lr_decay_method: {}'.format(m))


def split_global_batch(global_batch_size, world_size, rank):
  """Per-replica slice [begin, end) of the global batch; the reference requires divisibility
  (tf2/train.py:186-189: 'batch size must be divisible by number of replicas')."""
  if global_batch_size % world_size != 0:
    raise ValueError('batch size {} must be divisible by the number of replicas {}'.format(
        global_batch_size, world_size))
  per = global_batch_size // world_size
  return rank * per, (rank + 1) * per


def make_grad_all_reduce(process_group=None):
  """SUM all-reduce of the flat gradient arena (one RCCL call per step; 15.5 MB fp32 for D0).

  The reference clips the LOCAL gradient (per tensor, then global norm) before apply_gradients
  reduces it (train_lib.py:675-683), and the global-norm clip needs every local gradient, so the
  reduce cannot start before the backward pass has finished; see DESIGN.md section multi-GPU.
  """
  import torch.distributed as dist

  def reduce_fn(flat):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    return flat
  return reduce_fn


def ema_decay_dynamic(average_decay, num_updates):
  """TFA MovingAverage(dynamic_decay=True): min(decay, (1 + n) / (10 + n))."""
  return min(average_decay, (1.0 + num_updates) / (10.0 + num_updates))


def moving_normalizer_update(state, value, momentum):
  """The loss normalizer as a moving average (train_lib.py:519-531, config.positives_momentum > 0): Keras'
  moving_average_update on a variable that starts at 0.0 -- state <- state * momentum + value * (1 - momentum), with no
  zero-debiasing, so the first steps divide the loss by a small number exactly as the reference does.  `state` is a python
  float (returned) or a 0-d torch tensor (updated in place and returned: the device path, no host synchronisation)."""
  if torch.is_tensor(state):
    return state.mul_(momentum).add_(value, alpha=1.0 - momentum)
  return state * momentum + float(value) * (1.0 - momentum)


class EfficientDetNetTrain(efficientdet_net.EfficientDetNet):
  """EfficientDetNet plus the reference train_step.

  Data parallel: one process per GPU; pass ``process_group`` (torch.distributed, backend 'nccl'
  == RCCL over xGMI) and every replica's locally-clipped gradient is SUM all-reduced before the
  update, exactly the reference's MirroredStrategy semantics.
  """

  def __init__(self, *args, steps_per_epoch=1000, global_batch_size=None, process_group=None,
               use_dist=False, use_graph=False, sync_bn=False, overlap_grad_reduce=None, **kwargs):
    """use_graph: capture the whole step (forward, loss, backward, L2/clip, update: ~1600 kernel launches)
    into a hipGraph at the second call for a given batch shape and replay it afterwards; inputs are
    copied into static device buffers (input_buffers() exposes them for in-place filling), learning
    rate / EMA decay / loss normalizer travel through a small device vector.  With data parallelism the
    gradient all-reduce stays an eager RCCL call between two captured halves."""
    super().__init__(*args, **kwargs)
    self._check_training_options(self.config)
    # sync_bn: cross-replica BatchNorm statistics (the reference's --strategy=gpus / tpu BatchNorm classes,
    # utils.py:166-266): two small all-reduces per BatchNorm layer and step, eager launches only.
    self.sync_bn = sync_bn
    self.use_graph = use_graph and not sync_bn
    self._graph = None
    self.steps_per_epoch = steps_per_epoch
    self.global_batch_size = global_batch_size
    self.process_group = process_group
    self.use_dist = use_dist or process_group is not None
    self.one_graph_dp = os.environ.get('EDET_DP_ONE_GRAPH', '0') == '1'      # see _graph_step
    # The gradient all-reduce in buckets UNDER the backward pass (Engine.set_overlap_reduce) -- what north_star asks for.
    # Legal only with clip_gradients_norm = 0: the reference clips the local gradient by its global norm before the reduce
    # (train_lib.py:675-683), so with the default clip_gradients_norm = 10 the one flat reduce after the backward pass stays.
    if overlap_grad_reduce is None:
      overlap_grad_reduce = os.environ.get('EDET_DP_OVERLAP', '0') == '1'
    self.overlap_grad_reduce = bool(overlap_grad_reduce)
    if self.overlap_grad_reduce and self.config.clip_gradients_norm:
      raise ValueError('overlap_grad_reduce needs clip_gradients_norm=0 (the reference clips by the global norm of the '
                       'local gradient before the reduce, train_lib.py:675-683); got %r' % (self.config.clip_gradients_norm,))
    self.overlap_buckets = int(os.environ.get('EDET_DP_BUCKETS', '6'))
    self._lr_fn = None
    self.iterations = 0
    # positives_momentum > 0: the moving loss normalizer (train_lib.py:519-531).  A 0-d fp32 DEVICE tensor once the
    # engine exists (updated in place by both the graph and the eager step, no host synchronisation); a python float
    # only before that (a state restored into a net that has not stepped yet) or when the caller supplies the
    # normalizer of a step as a host float (labels['normalizer'])
    self._moving_normalizer = None
    # a list: every graph-mode data-parallel step appends (start, end) events recorded on the step's stream around the
    # gradient all-reduce (between the two captured graphs) -- its duration as the device sees it; None = off
    self.collective_timer = None

  def get_optimizer_state(self):
    """Optimizer slots, iteration count and -- with positives_momentum > 0 -- the moving loss normalizer.  (The reference
    creates that variable inside the loss call, train_lib.py:521-527, untracked by its checkpoints: a resumed reference
    run restarts the average at 0 and divides the first losses by ~(1 - m) * N.  Carrying it in the state avoids that
    spike; a state without the key restores to the reference's behaviour.)"""
    state = super().get_optimizer_state()
    if self._moving_normalizer is not None:
      state['moving_normalizer'] = float(self._moving_normalizer)
    return state

  def set_optimizer_state(self, state):
    """Optimizer slots + iteration count; the count also drives the learning-rate schedule and the dynamic EMA decay
    of the next train_step (optimizer.iterations in the reference, train_lib.py:37-173,193-197)."""
    super().set_optimizer_state(state)
    self.iterations = int(state['iterations'])
    if 'moving_normalizer' in state:
      value = float(state['moving_normalizer'])
      if torch.is_tensor(self._moving_normalizer):
        self._moving_normalizer.fill_(value)
      else:
        self._moving_normalizer = value
    elif torch.is_tensor(self._moving_normalizer):
      self._moving_normalizer.zero_()     # a state without the key: the reference's behaviour, the average restarts at 0
    else:
      self._moving_normalizer = None

  @staticmethod
  def _check_training_options(c):
    """The train step here is the reference's default one (train_lib.py:606-684 with the d0..d7x settings); options
    that would change its arithmetic and are not built raise instead of being ignored."""
    unsupported = []
    if getattr(c, 'iou_loss_type', None):
      unsupported.append('iou_loss_type=%r (BoxIouLoss, train_lib.py:440-466)' % c.iou_loss_type)
    if str(getattr(c, 'optimizer', 'sgd')).lower() not in ('sgd', 'adam'):
      unsupported.append("optimizer=%r (the reference has 'sgd' and 'adam', train_lib.py:180-188)" % c.optimizer)
    if unsupported:
      raise ValueError('training options that are not built: ' + '; '.join(unsupported))

  def _lr(self, batch):
    if self._lr_fn is None:
      p = self.config.as_dict()
      gbs = self.global_batch_size
      if gbs is None:
        # the reference scales by the GLOBAL batch (train_lib.py:41); per replica batch x number of replicas
        gbs = batch
        if self.use_dist:
          import torch.distributed as dist
          gbs = batch * dist.get_world_size(self.process_group)
      p['batch_size'] = gbs
      p['steps_per_epoch'] = self.steps_per_epoch
      self._lr_fn = learning_rate_schedule(p)
    return self._lr_fn(self.iterations)

  def _labels_to_device(self, labels, eng):
    out = {}
    for k, v in labels.items():
      if k == 'normalizer':
        out[k] = float(v)
        continue
      t = torch.as_tensor(v)
      if k.startswith('cls_targets'):
        t = t.to(device=eng.device, dtype=torch.int32)
      else:
        t = t.to(device=eng.device, dtype=torch.float32)
      out[k] = t.contiguous()
    return out

  # ---- hipGraph replay of the step ---------------------------------------------------------------
  def input_buffers(self):
    """(images, labels) static device buffers of the captured step (None before the first graph step):
    fill them in place and pass them to train_step to skip the staging copy."""
    g = self._graph
    return (g['images'], g['labels']) if g else None

  def _graph_step(self, eng, images, labels, lr, decay):
    g = self._graph
    if g is None or g['engine'] is not eng:
      g = self._graph = {'engine': eng, 'steps': 0, 'graphs': None,
                         'images': torch.empty_like(images),
                         'labels': {k: torch.empty_like(v) for k, v in labels.items() if torch.is_tensor(v)}}
    if images.data_ptr() != g['images'].data_ptr():
      g['images'].copy_(images, non_blocking=True)
    for k, buf in g['labels'].items():
      if labels[k].data_ptr() != buf.data_ptr():
        buf.copy_(labels[k], non_blocking=True)
    eng.set_hyper(lr, decay)
    if 'normalizer' in labels:                         # host value supplied by the caller
      norm = self._host_normalizer(float(labels['normalizer']))
      eng.hyper[2:3].copy_(torch.tensor([1.0 / norm], dtype=torch.float32), non_blocking=True)
    elif 'mean_num_positives' in g['labels']:
      self._device_normalizer(eng, g['labels']['mean_num_positives'])
    else:
      raise KeyError("labels need 'mean_num_positives' (dataloader.py:393) or a host 'normalizer'")
    glabels = dict(g['labels'])
    glabels['normalizer'] = 'device'
    reduce_fn = make_grad_all_reduce(self.process_group) if self.use_dist else None
    overlap = self.overlap_grad_reduce and reduce_fn is not None
    eng.set_overlap_reduce(reduce_fn if overlap else None, self.overlap_buckets)

    def body_a():
      eng.forward(g['images'], training=True)
      eng.loss_backward(glabels)           # (overlap: the bucketed all-reduce runs inside, on the communication stream)
      if overlap:
        eng.optimizer_apply(decay is not None, True)
        return
      eng.optimizer_local(reduce_fn is not None)
      if reduce_fn is None:
        eng.optimizer_apply(decay is not None, False)

    def body_b():
      eng.optimizer_apply(decay is not None, True)

    if g['steps'] == 0:
      # first step eager: allocates every buffer and runs the one-time kernel attribute setup
      body_a()
      if reduce_fn is not None and not overlap:
        reduce_fn(eng.grads_flat)
        body_b()
    else:
      if g['graphs'] is None:
        torch.cuda.synchronize()
        # the capture pass runs optimizer_apply's host bookkeeping once WITHOUT executing anything: keep the counters
        # where they were, the replay below accounts for the step
        counters = (eng.arena.version, eng.arena.step_count)
        ga = gb = None
        if overlap:
          # forward, backward with the bucketed collectives on their own stream (a parallel branch of the graph), update
          ga = torch.cuda.CUDAGraph()
          with torch.cuda.graph(ga, capture_error_mode='thread_local'):
            body_a()
          g['overlap'] = True
        elif reduce_fn is not None and self.one_graph_dp:
          # The collective captured INSIDE the step's graph (RCCL 2.26 supports stream capture): no host hop between the
          # two halves.  Opt-in (EDET_DP_ONE_GRAPH=1): exercised on the device at world size 1 only -- no multi-GPU node
          # was available to any round -- so the default stays the two-graph structure below; if the capture itself
          # raises, the step falls back to it.
          try:
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode='thread_local'):
              body_a()
              reduce_fn(eng.grads_flat)
              body_b()
          except Exception as e:      # noqa: BLE001 -- any capture failure: the robust structure
            import warnings
            warnings.warn('one-graph data-parallel capture failed (%s); using two graphs around an eager all-reduce' % (e,))
            ga = None
            torch.cuda.synchronize()
          eng.arena.version, eng.arena.step_count = counters
        if ga is None:
          ga = torch.cuda.CUDAGraph()
          # thread_local: other threads of the process (the RCCL watchdog) may touch the HIP runtime meanwhile
          with torch.cuda.graph(ga, capture_error_mode='thread_local'):
            body_a()
          if reduce_fn is not None and not overlap:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode='thread_local'):
              body_b()
          g['one_graph'] = False
        else:
          g['one_graph'] = True
        g['graphs'] = (ga, gb)
        eng.arena.version, eng.arena.step_count = counters
      ga, gb = g['graphs']
      eng.refresh_drop_masks()      # stochastic-depth draws live in static buffers the graph reads
      ga.replay()
      if gb is not None:
        timer = self.collective_timer
        if timer is not None:           # bench.py / the world-size-1 rehearsal: the collective's own duration
          e0 = torch.cuda.Event(enable_timing=True)
          e0.record()
        reduce_fn(eng.grads_flat)
        if timer is not None:
          e1 = torch.cuda.Event(enable_timing=True)
          e1.record()
          timer.append((e0, e1))
        gb.replay()
      eng.arena.version += 1        # what optimizer_apply does on the host when it is not replayed
      eng.arena.step_count += 1
    g['steps'] += 1

  def _ensure_engine(self, batch, height, width):
    eng = super()._ensure_engine(batch, height, width)
    expr = getattr(self.config, 'var_freeze_expr', None) or None
    if eng.arena.frozen_expr != expr:
      eng.arena.set_frozen(expr)         # tf2/train_lib.py:478-491: out of L2, gradients and updates (None: un-freeze)
    return eng

  def _positives_momentum(self):
    return float(getattr(self.config, 'positives_momentum', None) or 0.0)

  def _host_normalizer(self, value):
    """sum(mean_num_positives) + 1 as the reference uses it (train_lib.py:517-534): the moving average over the steps for
    positives_momentum > 0, the mean over the replicas for positives_momentum < 0, the value itself otherwise."""
    m = self._positives_momentum()
    if m > 0:
      if torch.is_tensor(self._moving_normalizer):      # one representation: the device scalar, once it exists
        return float(moving_normalizer_update(self._moving_normalizer, float(value), m))
      self._moving_normalizer = moving_normalizer_update(
          0.0 if self._moving_normalizer is None else float(self._moving_normalizer), value, m)
      return self._moving_normalizer
    if m < 0 and self.use_dist:
      import torch.distributed as dist
      t = torch.tensor([value], dtype=torch.float32)
      if dist.get_backend(self.process_group) == 'nccl':
        t = t.cuda()
      dist.all_reduce(t, group=self.process_group)
      return float(t.item()) / dist.get_world_size(self.process_group)
    return value

  def _device_normalizer(self, eng, mean_num_positives):
    """The same on the device (no host synchronisation; runs eagerly in front of the replayed graph): hyper[2] =
    1 / normalizer."""
    m = self._positives_momentum()
    if m == 0 or (m < 0 and not self.use_dist):
      eng.set_normalizer(mean_num_positives)
      return
    s = mean_num_positives.reshape(-1).float().sum() + 1.0
    if m > 0:
      if not torch.is_tensor(self._moving_normalizer):
        self._moving_normalizer = torch.full((), float(self._moving_normalizer or 0.0), dtype=torch.float32, device=s.device)
      s = moving_normalizer_update(self._moving_normalizer, s, m)
    else:
      import torch.distributed as dist
      s = s.clone()
      dist.all_reduce(s, group=self.process_group)
      s = s / dist.get_world_size(self.process_group)
    torch.reciprocal(s, out=eng.hyper[2])

  def train_step(self, data, sync_loss=True):
    images, labels = data
    b, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
    eng = self._ensure_engine(b, h, w)
    if self.sync_bn and self.use_dist and eng.sync_bn is None:
      import torch.distributed as dist
      eng.sync_bn = (make_grad_all_reduce(self.process_group), dist.get_world_size(self.process_group))
    lr = self._lr(b)
    if self.use_graph:
      decay = None
      if self.config.moving_average_decay:
        decay = ema_decay_dynamic(self.config.moving_average_decay, self.iterations)
      self._graph_step(eng, self._to_device_images(images, eng), self._labels_to_device(labels, eng), lr, decay)
      self.iterations += 1
      if not sync_loss:
        return {'learning_rate': lr}
      vals = eng.loss_values()
      vals['learning_rate'] = lr
      return vals
    reduce_fn = make_grad_all_reduce(self.process_group) if self.use_dist else None
    eng.set_overlap_reduce(reduce_fn if self.overlap_grad_reduce else None, self.overlap_buckets)
    eng.forward(self._to_device_images(images, eng), training=True)
    dlabels = self._labels_to_device(labels, eng)
    if self._positives_momentum() != 0:
      if 'normalizer' in dlabels:
        # the caller supplied this step's sum(mean_num_positives) + 1 as a host float: moving average / replica mean on the host
        dlabels['normalizer'] = self._host_normalizer(dlabels['normalizer'])
      else:
        # on the device, as the captured step does it (no .item(): the eager step does not wait for the host either)
        self._device_normalizer(eng, dlabels['mean_num_positives'])
        dlabels['normalizer'] = 'device'
    eng.loss_backward(dlabels)
    decay = None
    if self.config.moving_average_decay:
      decay = ema_decay_dynamic(self.config.moving_average_decay, self.iterations)
    eng.optimizer_step(lr, decay, reduce_fn)
    self.iterations += 1
    if not sync_loss:
      return {'learning_rate': lr}
    vals = eng.loss_values()
    vals['learning_rate'] = lr
    return vals
