"""Config parser behaviour, mirroring the reference's hparams_config_test.py:27-83, plus golden
tables generated from the reference's own module (tests/golden/make_golden.py)."""
import json
import os
import tempfile

import pytest
import yaml

from automl_amd import hparams_config

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'reference_tables.json')


def test_config_override():
  c = hparams_config.Config({'a': 1, 'b': 2})
  assert c.as_dict() == {'a': 1, 'b': 2}
  c.update({'a': 10})
  assert c.as_dict() == {'a': 10, 'b': 2}
  c.b = 20
  assert c.as_dict() == {'a': 10, 'b': 20}
  c.override('a=true,b=ss')
  assert c.as_dict() == {'a': True, 'b': 'ss'}
  c.override('a=100,,,b=2.3,')  # extra ',' is fine.
  assert c.as_dict() == {'a': 100, 'b': 2.3}
  c.override('a=2x3,b=50')  # special format for image size.
  assert c.as_dict() == {'a': '2x3', 'b': 50}
  with pytest.raises(ValueError):
    c.override('a=true,invalid_string')


def test_config_yaml():
  tmpdir = tempfile.gettempdir()
  p1 = os.path.join(tmpdir, 'edet_x.yaml')
  with open(p1, 'w') as f:
    f.write("""
        x: 2
        y:
          z: 'test'
      """)
  c = hparams_config.Config(dict(x=234, y=2342))
  c.override(p1)
  assert c.as_dict() == {'x': 2, 'y': {'z': 'test'}}
  p2 = os.path.join(tmpdir, 'edet_y.yaml')
  c.save_to_yaml(p2)
  with open(p2, 'r') as f:
    assert yaml.load(f, Loader=yaml.FullLoader) == {'x': 2, 'y': {'z': 'test'}}


def test_config_override_recursive():
  c = hparams_config.Config({'x': 1})
  c.override('y.y0=2,y.y1=3', allow_new_keys=True)
  assert c.as_dict() == {'x': 1, 'y': {'y0': 2, 'y1': 3}}
  c.update({'y': {'y0': 5, 'y1': {'y11': 100}}})
  assert c.as_dict() == {'x': 1, 'y': {'y0': 5, 'y1': {'y11': 100}}}
  assert c.y.y1.y11 == 100


def test_config_override_list():
  c = hparams_config.Config({'x': [1.0, 2.0]})
  c.override('x=3.0*4.0*5.0')
  assert c.as_dict() == {'x': [3.0, 4.0, 5.0]}


def test_unknown_key_rejected():
  c = hparams_config.get_efficientdet_config('efficientdet-d0')
  with pytest.raises(KeyError):
    c.override('no_such_key=1')
  with pytest.raises(ValueError):
    hparams_config.get_efficientdet_config('efficientdet-d9')
  with pytest.raises(ValueError):
    hparams_config.get_detection_config('retinanet')


def test_model_tables_equal_reference():
  """Every named model's full config dict equals the dict produced by the reference module."""
  with open(GOLDEN) as f:
    gold = json.load(f)
  assert len(gold['models']) == 15
  for name, want in gold['models'].items():
    got = json.loads(json.dumps(hparams_config.get_efficientdet_config(name).as_dict()))
    assert got == want, name


def test_override_strings_equal_reference():
  with open(GOLDEN) as f:
    gold = json.load(f)
  for case in gold['override_cases']:
    c = hparams_config.get_efficientdet_config('efficientdet-d0')
    c.override(case['str'])
    assert json.loads(json.dumps(c.as_dict())) == case['result'], case['str']


def test_unbuilt_options_raise_instead_of_being_ignored():
  """Options of the reference that would change the arithmetic of the network / train step and are not built must
  fail loudly (constructing the host objects needs no GPU)."""
  import pytest
  from automl_amd import efficientdet_net, train_lib
  for override in ('iou_loss_type=ciou', 'optimizer=rmsprop'):
    config = hparams_config.get_efficientdet_config('efficientdet-d0')
    config.override(override)
    with pytest.raises(ValueError, match='not built'):
      train_lib.EfficientDetNetTrain(config=config)
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('optimizer=adam')         # tf.keras.optimizers.Adam(lr, beta_1=momentum): built in r06 (edet_opt_adam_ema)
  train_lib.EfficientDetNetTrain(config=config)
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('survival_prob=0.8')      # stochastic depth in the towers: built since r04 (Engine._head_level)
  efficientdet_net.EfficientDetNet(config=config)
  train_lib.EfficientDetNetTrain(config=config)
  train_lib.EfficientDetNetTrain(config=hparams_config.get_efficientdet_config('efficientdet-d0'))   # defaults pass


def test_positives_momentum_is_the_references_moving_normalizer():
  """config.positives_momentum > 0 (tf2/train_lib.py:519-531): the loss normalizer sum(mean_num_positives) + 1 goes through
  Keras' moving_average_update on a variable that starts at 0.0 -- v <- v * m + x * (1 - m), no zero-debiasing.  The host
  path (python floats) and the device path (the same torch code on a 0-d tensor, in place) must both follow the closed
  form v_t = (1 - m) * sum_k m^(t-k) x_k; constructing the host objects needs no GPU."""
  import numpy as np
  import torch
  from automl_amd import train_lib
  m = 0.9
  xs = [101.0, 57.0, 230.5, 1.0, 88.0]
  want, v = [], 0.0
  for x in xs:
    v = v - (v - x) * (1.0 - m)          # Keras: variable -= (variable - value) * (1 - momentum)
    want.append(v)
  closed = [(1 - m) * sum(m ** (t - k) * xs[k] for k in range(t + 1)) for t in range(len(xs))]
  np.testing.assert_allclose(want, closed, rtol=1e-12)
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('positives_momentum=%r' % m)
  net = train_lib.EfficientDetNetTrain(config=config)
  got = [net._host_normalizer(x) for x in xs]
  np.testing.assert_allclose(got, want, rtol=1e-12)
  state = torch.zeros((), dtype=torch.float32)
  dev = [float(train_lib.moving_normalizer_update(state, torch.tensor(x), m)) for x in xs]
  np.testing.assert_allclose(dev, want, rtol=1e-6)
  # off by default: the value itself
  net0 = train_lib.EfficientDetNetTrain(config=hparams_config.get_efficientdet_config('efficientdet-d0'))
  assert net0._host_normalizer(101.0) == 101.0 and net0._moving_normalizer is None
  # one representation once the device scalar exists: a host-supplied normalizer updates THAT tensor (a net that ran graph
  # steps and is then handed a host float continues the same average), and the state of the average travels with the
  # optimizer state (the reference's variable is untracked: a resumed run would restart it at 0)
  net2 = train_lib.EfficientDetNetTrain(config=config)
  net2._moving_normalizer = torch.zeros((), dtype=torch.float32)
  got2 = [net2._host_normalizer(x) for x in xs]
  np.testing.assert_allclose(got2, want, rtol=1e-6)
  assert torch.is_tensor(net2._moving_normalizer) and abs(float(net2._moving_normalizer) - want[-1]) < 1e-3

  class _Arena(object):      # (the optimizer slots need a GPU engine; the normalizer part of the state does not)
    def get_optimizer_state(self):
      return {'iterations': 7}

    def set_optimizer_state(self, state):
      self.restored = dict(state)

  class _Engine(object):
    arena = _Arena()
  net2.engine = _Engine()
  state = net2.get_optimizer_state()
  assert abs(state['moving_normalizer'] - want[-1]) < 1e-3 and state['iterations'] == 7
  net3 = train_lib.EfficientDetNetTrain(config=config)
  net3.engine = _Engine()
  net3.set_optimizer_state(state)
  assert net3.iterations == 7 and abs(net3._moving_normalizer - want[-1]) < 1e-3      # a float until the engine steps
  nxt = net3._host_normalizer(50.0)
  assert abs(nxt - (want[-1] * m + 50.0 * (1 - m))) < 1e-3
  net2.set_optimizer_state({'iterations': 9, 'moving_normalizer': 12.5})             # into the existing device scalar
  assert torch.is_tensor(net2._moving_normalizer) and float(net2._moving_normalizer) == 12.5
  net0.engine = _Engine()
  assert 'moving_normalizer' not in net0.get_optimizer_state()


def test_var_freeze_expr_in_the_arena_and_in_the_oracle():
  """config.var_freeze_expr (tf2/train_lib.py:478-491): re.match on the variable name with TensorFlow's ':0'.  The
  parameter arena (built on the CPU here: no kernel is launched) clears the L2 flag of the frozen variables and records
  their element ranges -- the reference's finetuning expression freezes 409 of d0's 493 trainable variables, one contiguous
  range -- and the oracle's train step leaves them out of L2, gradients and update."""
  import numpy as np
  import torch
  from automl_amd import engine, netspec
  from oracle import efficientdet_oracle as orc
  from oracle.problems import perturbed_params
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=64,var_freeze_expr=(efficientnet|fpn_cells|resample_p6)')
  spec = netspec.NetSpec(config)
  vals = perturbed_params(config, 3)
  arena = engine.ParamArena(spec, 'cpu', vals)
  flags_before = arena.seg_flags.clone()
  l2_before = int((arena.seg_flags & 1).sum())
  arena.velocity.fill_(1.0)
  frozen = arena.set_frozen(config.var_freeze_expr)
  assert len(frozen) == 409 and all(n.startswith(('efficientnet', 'fpn_cells', 'resample_p6')) for n in frozen)
  assert len(arena.frozen_ranges) == 1 and arena.frozen_ranges[0][0] == 0
  end = arena.frozen_ranges[0][1]
  assert end == max(arena.offsets[n][0] + arena.offsets[n][1] for n in frozen)
  assert float(arena.velocity[:end].abs().max()) == 0.0 and float(arena.velocity[end:].min()) == 1.0
  kept = [n for n in arena.seg_names if n not in frozen]
  assert int((arena.seg_flags & 1).sum()) == sum(1 for n in kept if orc.is_l2_regularised(n)) < l2_before
  assert int((arena.seg_flags & 2).ne(0).sum()) == len(frozen)        # EDET_SEG_FROZEN: the optimizer kernels skip these
  # another expression starts from the arena's own flags: what no longer matches is trainable again, with
  # its L2 flag; an empty expression un-freezes everything
  again = arena.set_frozen('(efficientnet)')
  assert 0 < len(again) < len(frozen) and int((arena.seg_flags & 2).ne(0).sum()) == len(again)
  assert int((arena.seg_flags & 1).sum()) == sum(1 for n in arena.seg_names if n not in again and orc.is_l2_regularised(n))
  assert arena.set_frozen(None) == [] and torch.equal(arena.seg_flags, flags_before) and arena.frozen_ranges == []
  frozen = arena.set_frozen(config.var_freeze_expr)
  # the oracle's step
  rng = np.random.default_rng(5)
  images = torch.from_numpy(rng.standard_normal((1, 64, 64, 3)).astype(np.float32))
  from tests.test_gpu_network import make_labels          # (label maps only: nothing of that module touches a GPU here)
  labels = {k: torch.from_numpy(v) for k, v in make_labels(config, 1, 64, 7).items()}
  oracle = orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
  with torch.no_grad():
    oracle.forward(images, False)
  out, grads = orc.train_step(oracle, images, labels, {}, 0.05, None)
  assert sorted(grads) == sorted(kept)
  want_l2 = config.weight_decay * sum(float((vals[n].astype(np.float64) ** 2).sum()) / 2 for n in kept if orc.is_l2_regularised(n))
  assert abs(out['reg_l2_loss'] - want_l2) <= 1e-5 * want_l2
  P = oracle.params()
  assert all(np.array_equal(P[n].detach().numpy(), vals[n]) for n in frozen)
  # (the per-level BatchNorm variables of the 1x1 / 2x2 levels of a 64-pixel image see no gradient: most, not all, move)
  assert sum(1 for n in kept if not np.array_equal(P[n].detach().numpy(), vals[n])) >= len(kept) // 2
  assert not np.array_equal(P['class_net/class-predict/bias'].detach().numpy(), vals['class_net/class-predict/bias'])


def test_config_pickle_and_copy_round_trip():
  """Config travels through pickle / copy (torch.multiprocessing spawn arguments, DataLoader workers): the slot-based
  table needs an explicit reduce."""
  import copy
  import pickle
  c = hparams_config.get_efficientdet_config('efficientdet-d1')
  c.override('image_size=640,nms_configs.sigma=0.7')
  for clone in (pickle.loads(pickle.dumps(c)), copy.copy(c), copy.deepcopy(c)):
    assert isinstance(clone, hparams_config.Config) and clone.as_dict() == c.as_dict()
    assert isinstance(clone.nms_configs, hparams_config.Config)
    clone.nms_configs.sigma = 0.1
    assert c.nms_configs.sigma == 0.7
