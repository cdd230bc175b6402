"""Checkpoint interchange with the reference: mirror of ``efficientdet/tf2/util_keras.py:67-203``.

``restore_ckpt(model, ckpt_path_or_file, ema_decay, skip_mismatch, exclude_layers)`` loads the reference's checkpoints
into the MI355X model classes (``EfficientDetNet`` / ``EfficientDetModel`` / ``EfficientDetNetTrain``,
``effnetv2_model.EffNetV2Model``) -- the published ``efficientdet-d*.tar.gz`` archives are name-based TF1 checkpoints whose
keys are the variable names of SURVEY.md appendix B plus ``<name>/ExponentialMovingAverage`` shadows; checkpoints written by
the reference's Keras training (``tf2/train.py``) are object-based.  ``save_ckpt`` writes the model back in the name-based
layout, so weights trained here can be restored by the reference's ``restore_ckpt`` / ``tf.train.load_checkpoint``.
The file format lives in ``automl_amd/tf_checkpoint.py`` (no TensorFlow).

Semantics followed line by line (util_keras.py):
  * ``'_'`` loads nothing (:126-128); a directory means its latest checkpoint (:129-130);
  * name-based (:159-203): every EMA variable (trainable variables + BatchNorm moving statistics, :67-80) is looked up
    under its own name, and -- when ``ema_decay > 0`` -- its shadow under ``<name>/ExponentialMovingAverage``
    (``tf.train.ExponentialMovingAverage.average_name``); with an optimizer on the model the shadow goes to the optimizer's
    ``average`` slot, without one it OVERWRITES the variable (the dict is walked in insertion order: plain names first);
    a key that is missing or has another shape is skipped with a warning under ``skip_mismatch`` and raises
    (KeyError / ValueError with the reference's messages) otherwise;
  * object-based (:133-157): the top-level attributes named by the checkpoint keys (minus ``exclude_layers``) are
    restored; a checkpoint that matches nothing is tried as an EfficientDetNetTrainHub checkpoint
    (``load_from_hub_checkpoint``, :83-105, key mapping HUB_CPT_NAME).  Variables are matched through the
    ``full_name`` the object graph records for each of them, which is the variable name the model classes here use.
"""
import collections
import logging
import os

import numpy as np

from automl_amd import tf_checkpoint

# Prefix variable name mapping from tf2 model to the hub module checkpoint (util_keras.py:24-26).
HUB_CPT_NAME = collections.OrderedDict([('class_net/class-predict/', 'classes'),
                                        ('box_net/box-predict/', 'boxes'),
                                        ('', 'base_model')])
EMA_SUFFIX = '/ExponentialMovingAverage'


def average_name(name):
  """tf.train.ExponentialMovingAverage.average_name for a variable called ``name`` (no ':0')."""
  return name + EMA_SUFFIX


def model_variables(model):
  """[(name, shape, trainable)] of the model's variables, in creation order, without building device buffers."""
  spec = getattr(model, 'spec', None)
  if spec is None:
    eng = getattr(model, 'engine', None)
    if eng is not None:
      spec = eng.spec
  if spec is None:
    from automl_amd import netspec
    spec = netspec.NetSpec(model.config)
  return [(p.name, tuple(p.shape), p.trainable) for p in spec.params]


def get_ema_vars(model):
  """Names of the variables that have an EMA shadow: the trainable ones plus the BatchNorm moving statistics
  (util_keras.py:67-80)."""
  out = collections.OrderedDict()
  for name, _, trainable in model_variables(model):
    if trainable or 'moving_mean' in name or 'moving_variance' in name:
      out[name] = True
  return list(out)


def _has_optimizer(model):
  """True for the training model (its MovingAverage optimizer holds the shadows in 'average' slots)."""
  return hasattr(model, 'train_step')


def _assign(model, values, ema_values):
  if values:
    model.set_weights(values)
  if ema_values:
    setter = getattr(model, 'set_ema_weights', None)
    if setter is None:
      raise ValueError('the model has no EMA slots to restore into')
    setter(ema_values)


def load_from_hub_checkpoint(model, ckpt_path_or_file):
  """Loads EfficientDetNet weights from an EfficientDetNetTrainHub checkpoint (util_keras.py:83-105)."""

  def hub_key(variable_name):
    """Checkpoint key of a model variable (name WITH its ':0' suffix, as the reference passes var.name): the first
    matching prefix of HUB_CPT_NAME is cut off, '/' becomes '.S' and the hub attribute name goes in front; only the
    catch-all 'base_model' entry (empty prefix) keeps the ':0'."""
    for prefix, hub_attr in HUB_CPT_NAME.items():
      if not variable_name.startswith(prefix):
        continue
      tail = variable_name[len(prefix):].replace('/', '.S')
      if prefix:
        tail = tail.replace(':0', '')
      return '%s/%s/.ATTRIBUTES/VARIABLE_VALUE' % (hub_attr, tail)
    raise KeyError(variable_name)

  reader = tf_checkpoint.load_checkpoint(ckpt_path_or_file)
  values = {}
  for name, shape, _ in model_variables(model):
    # the reference passes var.name, which ends in ':0'; the base_model branch keeps that suffix in the key
    key = hub_key(name + ':0')
    v = reader.get_tensor(key)
    values[name] = np.asarray(v, np.float32).reshape(shape)
  _assign(model, values, None)


def restore_ckpt(model, ckpt_path_or_file, ema_decay=0.9998, skip_mismatch=True, exclude_layers=None):
  """Restore variables from a given checkpoint (util_keras.py:108-203).

  Args:
    model: EfficientDetNet / EfficientDetModel / EfficientDetNetTrain (or any object with ``config`` or ``spec``,
      ``set_weights(dict)`` and, for the training model, ``set_ema_weights(dict)``).
    ckpt_path_or_file: checkpoint prefix, or a directory holding a ``checkpoint`` state file; '_' loads nothing.
    ema_decay: ema decay rate. If None or zero or negative value, disable ema.
    skip_mismatch: whether to skip variables if shape mismatch, only works with tf1 (name-based) checkpoints.
    exclude_layers: top-level attributes (backbone, resample_layers, fpn_cells, class_net, box_net) whose variables are
      left alone, only works with tf2 (object-based) checkpoints.

  Raises:
    KeyError / ValueError: a variable is missing / has another shape and skip_mismatch is False.

  Limitation (object-based checkpoints): the reference builds ``tf.train.Checkpoint(**{key: getattr(model, key)})`` and
  TensorFlow matches variables by walking the object graph's attribute paths; here a saved variable is matched through
  the ``full_name`` its SerializedTensor carries (= the Keras variable name, which is what this code base addresses
  variables by), restricted to the top-level attributes present.  A checkpoint whose variables were saved with an
  empty / different ``full_name`` (e.g. a model built under another name scope) does not match and falls through to
  the hub loader, exactly as a trivial match does in the reference.
  """
  if ckpt_path_or_file == '_':
    logging.info('Running test: do not load any ckpt.')
    return
  if os.path.isdir(ckpt_path_or_file):
    latest = tf_checkpoint.latest_checkpoint(ckpt_path_or_file)
    if latest is None:
      raise FileNotFoundError('no checkpoint found in directory %s' % ckpt_path_or_file)
    ckpt_path_or_file = latest
  reader = tf_checkpoint.CheckpointReader(ckpt_path_or_file)
  variables = model_variables(model)
  shapes = {name: shape for name, shape, _ in variables}

  if reader.has_tensor(tf_checkpoint.OBJECT_GRAPH_KEY):
    by_name, slots, top = tf_checkpoint.object_graph_variables(reader)
    keys = set(top.values())
    keys.discard(tf_checkpoint.OBJECT_GRAPH_KEY)
    if exclude_layers:
      keys = keys.difference(set(exclude_layers))
    values, ema_values = {}, {}
    for name, shape, _ in variables:
      key = by_name.get(name)
      if key is None or top.get(key) not in keys:      # not saved, or under an excluded top-level attribute
        continue
      v = reader.get_tensor(key)
      if tuple(v.shape) != shape:
        # tf.train.Checkpoint.restore raises on incompatible shapes; util_keras.py:136 comments on exactly that
        raise ValueError('Shape mismatch: %s, expected %s, but got %s' % (name, str(shape), str(tuple(v.shape))))
      values[name] = np.asarray(v, np.float32)
      skey = slots.get((name, 'average'))
      if skey is not None and 'optimizer' not in (exclude_layers or ()) and _has_optimizer(model):
        ema_values[name] = np.asarray(reader.get_tensor(skey), np.float32)
    if values:                                   # status.assert_nontrivial_match()
      _assign(model, values, ema_values)
      return
    load_from_hub_checkpoint(model, ckpt_path_or_file)
    return

  ema_vars = get_ema_vars(model)
  # insertion order matters: the plain name first, the shadow second (it overwrites when there is no optimizer)
  var_dict = collections.OrderedDict((name, ('var', name)) for name in ema_vars)
  if ema_decay is not None and ema_decay > 0:
    target = 'slot' if _has_optimizer(model) else 'var'
    for name in ema_vars:
      var_dict[average_name(name)] = (target, name)
  for name, _, _ in variables:
    if name not in var_dict:
      var_dict[name] = ('var', name)
  var_shape_map = reader.get_variable_to_shape_map()
  values, ema_values = collections.OrderedDict(), collections.OrderedDict()
  for key, (kind, name) in var_dict.items():
    if key in var_shape_map:
      if tuple(var_shape_map[key]) != shapes[name]:
        msg = 'Shape mismatch: %s, expected %s, but got %s' % (key, str(shapes[name]), str(tuple(var_shape_map[key])))
        if skip_mismatch:
          logging.warning(msg)
        else:
          raise ValueError(msg)
      else:
        v = np.asarray(reader.get_tensor(key), np.float32)
        if kind == 'var':
          values[name] = v
        else:
          ema_values[name] = v
    else:
      msg = 'Not found %s in %s' % (key, ckpt_path_or_file)
      if skip_mismatch:
        logging.warning(msg)
      else:
        raise KeyError(msg)
  if ema_values:
    # the training model keeps shadows of the trainable variables only (TFA MovingAverage averages what the
    # optimizer updates, train_lib.py:193-197); shadows of the moving statistics have no slot to go to
    trainable = {name for name, _, tr in variables if tr}
    ema_values = {k: v for k, v in ema_values.items() if k in trainable}
  _assign(model, values, ema_values)


def save_ckpt(model, ckpt_prefix, ema=True, global_step=None):
  """Writes the model as a name-based checkpoint in the layout of the published EfficientDet archives: every variable
  under its name and, when ``ema`` and the model carries shadows, ``<name>/ExponentialMovingAverage`` next to it (for
  the moving statistics, which have no shadow here, the statistic itself -- what a decay-0 shadow would hold);
  ``global_step`` (int64 scalar) when given.  Also updates the directory's ``checkpoint`` state file."""
  weights = model.get_weights()
  tensors = {name: np.asarray(v, np.float32) for name, v in weights.items()}
  if ema:
    shadows = {}
    getter = getattr(model, 'get_ema_weights', None)
    if getter is not None:
      shadows = getter()
    for name in get_ema_vars(model):
      tensors[average_name(name)] = np.asarray(shadows.get(name, weights[name]), np.float32)
  if global_step is not None:
    tensors['global_step'] = np.asarray(global_step, np.int64)
  tf_checkpoint.write_checkpoint(ckpt_prefix, tensors)
  tf_checkpoint.update_checkpoint_state(os.path.dirname(os.path.abspath(ckpt_prefix)), os.path.abspath(ckpt_prefix))
  return ckpt_prefix
