"""A recorded step plan replayed through the network-level C ABI (include/edet_net.h: edet_create / edet_forward /
edet_train_step) equals the Python host's own passes BIT FOR BIT: in process through ctypes (eager and as a captured
hipGraph), and from a C99 host program with no interpreter (tests/c_host/edet_host.c, built here with gcc).  Reference
interfaces: efficientdet/tf2/efficientdet_keras.py:790-799, 893-915; efficientdet/tf2/train_lib.py:606-684."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from automl_amd import hparams_config, net_c, plan, train_lib
from tests.test_gpu_network import make_labels, perturbed_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, DECAY = 0.02, 0.9
RAW_HW = (300, 400)      # raw images of the `detect` program (resized into the top-left corner of SIZE x SIZE)
SIZE = 256      # level 3 is 32 x 32: the heads run as two chains (fork / join events in the plan), levels 4-7 on the side stream


@pytest.fixture(scope='module')
def recorded(tmp_path_factory):
  """d0 at 256 x 256, two images, bf16: the plan, what the Python host computed while recording, and the engine (to run
  further Python steps against further replayed ones)."""
  d = tmp_path_factory.mktemp('plan')
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=perturbed_params(config, 11), seed=5)
  rng = np.random.default_rng(97)
  images = torch.from_numpy(rng.standard_normal((2, SIZE, SIZE, 3)).astype(np.float32))
  labels = make_labels(config, 2, SIZE, 101)
  path = str(d / 'd0_256_b2.plan')
  raw = torch.from_numpy(rng.integers(0, 256, (2,) + RAW_HW + (3,), dtype=np.uint8))
  summary, expected = plan.record_network(net, images, labels, path, learning_rate=LR, ema_decay=DECAY,
                                          detect_raw_hw=RAW_HW, raw_images=raw)
  return {'path': path, 'summary': summary, 'expected': expected, 'net': net, 'images': images, 'labels': labels,
          'dir': str(d), 'config': config}


def _logits_equal(cnet, expected, config):
  for level in range(config.min_level, config.max_level + 1):
    for kind in ('cls', 'box'):
      name = '%s_outputs_%d' % (kind, level)
      got, want = cnet.read(name), expected[name]
      assert got.shape == want.shape, name
      # compare the logits proper: the first `channels` elements of every pixel (the padding columns of a row are not
      # part of the result and are not written by every kernel)
      eb, ld, ch = cnet.prop(name + '.elem_bytes'), cnet.prop(name + '.ld'), cnet.prop(name + '.channels')
      g = got.reshape(-1, ld * eb)[:, :ch * eb]
      w = want.reshape(-1, ld * eb)[:, :ch * eb]
      assert np.array_equal(g, w), '%s: %d bytes differ' % (name, int((g != w).sum()))


DET_DTYPES = {'boxes': np.float32, 'scores': np.float32, 'classes': np.float32, 'valid_len': np.int32}


def _detections_equal(cnet, expected):
  for k, dt in DET_DTYPES.items():
    got = cnet.read('detections.' + k).view(dt)
    want = expected['detections.' + k].reshape(-1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), 'detections.%s differs' % k
  assert int(expected['detections.valid_len'].min()) >= 0


def test_detect_program_equals_the_python_model(tmp_path):
  """An inference-only plan (no training step: the variables never change): the recorded `detect` pass and its replay
  equal the Python host's own EfficientDetModel path -- preprocess_infer, the network, postprocess_global
  (efficientdet_keras.py:920-1000) -- bit for bit: boxes in raw-image pixels, scores, classes, valid_len."""
  from automl_amd import efficientdet_net, postprocess, preprocess
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=%d' % SIZE)
  net = efficientdet_net.EfficientDetNet(config=config, dtype='bf16', params=perturbed_params(config, 11), seed=5)
  rng = np.random.default_rng(7)
  raw = torch.from_numpy(rng.integers(0, 256, (2,) + RAW_HW + (3,), dtype=np.uint8))
  images, scales = preprocess.preprocess_infer(raw, SIZE, config.mean_rgb, config.stddev_rgb, dtype=torch.bfloat16)
  cls_out, box_out = net(images, training=False)
  want = postprocess.postprocess_global(config.as_dict(), cls_out, box_out, scales)
  want = [t.cpu().numpy().copy() for t in want]
  assert int(want[3].max()) > 0, 'the problem must produce detections'
  path = str(tmp_path / 'd0_detect.plan')
  summary, expected = plan.record_network(net, images, None, path, detect_raw_hw=RAW_HW, raw_images=raw)
  assert set(summary['programs']) == {'forward', 'detect'}
  for k, w in zip(('boxes', 'scores', 'classes', 'valid_len'), want):
    assert np.array_equal(expected['detections.' + k].view(np.uint32).reshape(-1), w.view(np.uint32).reshape(-1)), k
  cnet = net_c.CNet(path)
  try:
    assert cnet.prop('raw_height') == RAW_HW[0] and cnet.prop('max_output_size') == 100 and not cnet.has_program('train_step')
    cnet.use_graph(True)
    st = torch.cuda.Stream()
    for _ in range(3):
      cnet.detect(st.cuda_stream)
      st.synchronize()
      _detections_equal(cnet, expected)
    # other raw images through the same plan: the host writes "raw_images", the replayed graph reads it
    raw2 = torch.from_numpy(rng.integers(0, 256, (2,) + RAW_HW + (3,), dtype=np.uint8))
    cnet.write('raw_images', raw2.numpy())
    cnet.detect(st.cuda_stream)
    st.synchronize()
    images2, scales2 = preprocess.preprocess_infer(raw2, SIZE, config.mean_rgb, config.stddev_rgb, dtype=torch.bfloat16)
    c2, b2 = net(images2, training=False)
    want2 = postprocess.postprocess_global(config.as_dict(), c2, b2, scales2)
    for k, w in zip(('boxes', 'scores', 'classes', 'valid_len'), want2):
      got = cnet.read('detections.' + k).view(np.uint32)
      assert np.array_equal(got, w.cpu().numpy().view(np.uint32).reshape(-1)), k
  finally:
    cnet.close()


def test_plan_summary(recorded):
  s = recorded['summary']
  assert set(s['programs']) == {'forward', 'detect', 'train_step'}
  assert s['programs']['train_step'] > s['programs']['forward'] > 100
  assert s['streams'] == 2 and s['events'] >= 2      # the two head chains: one fork + one join per pass at least
  got = plan.read_plan(recorded['path'])
  assert any(o[0] == 'allreduce' for o in got['ops']['train_step'])
  assert {'images', 'params', 'ema', 'hyper', 'loss_sums', 'mean_num_positives', 'cls_outputs_3'} <= set(got['names'])


def test_replay_in_process_equals_the_python_host(recorded):
  cnet = net_c.CNet(recorded['path'])
  try:
    assert cnet.prop('batch') == 2 and cnet.prop('height') == SIZE and cnet.has_program('train_step')
    cnet.forward()
    _logits_equal(cnet, recorded['expected'], recorded['config'])
    # as a captured graph on a stream of its own: eager, capture + launch, replay -- the same bits every time
    cnet.use_graph(True)
    st = torch.cuda.Stream()
    for _ in range(3):
      cnet.forward(st.cuda_stream)
      st.synchronize()
      _logits_equal(cnet, recorded['expected'], recorded['config'])
    # raw images -> detections, eager then captured then replayed
    for _ in range(3):
      cnet.detect(st.cuda_stream)
      st.synchronize()
      _detections_equal(cnet, recorded['expected'])
    cnet.use_graph(False)
    cnet.train_step(LR, DECAY)
    for k in ('params', 'ema', 'velocity', 'bn_state', 'loss_sums'):
      got = cnet.read(k).view(np.float32)
      want = recorded['expected'][k].reshape(-1)
      assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), '%s: %d of %d elements differ' % (
          k, int((got.view(np.uint32) != want.view(np.uint32)).sum()), want.size)
  finally:
    cnet.close()


def test_replayed_steps_track_python_steps_and_the_exchange_callback_runs(recorded):
  """Three further steps: the Python engine continues from its state after the recorded step, the replay (graph mode, a
  gradient-exchange callback installed: world size 1 = identity, but it must be called once per step with the arena)
  starts from the plan's initial state and so runs one step more; variables and losses bit-equal after every step."""
  net = recorded['net']
  eng = net._ensure_engine(2, SIZE, SIZE)
  images = net._to_device_images(recorded['images'], eng)
  dl = net._labels_to_device(recorded['labels'], eng)
  cnet = net_c.CNet(recorded['path'])
  seen = []
  try:
    cnet.dp_init(lambda buf, count, stream: seen.append((buf, count)) or 0)
    cnet.use_graph(True)
    st = torch.cuda.Stream()
    cnet.train_step(LR, DECAY, st.cuda_stream)       # = the recorded step
    st.synchronize()
    assert np.array_equal(cnet.read('params').view(np.uint32), recorded['expected']['params'].view(np.uint32).reshape(-1))
    for step in range(3):
      lr = LR * (1 + step)
      plan.train_pass(eng, images, dl, lr, DECAY)
      cnet.train_step(lr, DECAY, st.cuda_stream)
      st.synchronize()
      torch.cuda.synchronize()
      for k, t in (('params', eng.params_flat), ('ema', eng.ema), ('bn_state', eng.state_flat), ('loss_sums', eng.loss_sums)):
        got = cnet.read(k).view(np.uint32)
        want = t.detach().cpu().numpy().view(np.uint32).reshape(-1)
        assert np.array_equal(got, want), 'step %d, %s: %d elements differ' % (step, k, int((got != want).sum()))
    # eager run: one call; capture run: one call (captured as part of the graph); replays: none (inside the graph)
    assert len(seen) >= 2 and all(c == cnet.prop('num_train_elems') for _, c in seen)
    assert all(b == cnet.buffer('params')[0] or b for b, _ in seen)
  finally:
    cnet.close()


def test_c_host_without_an_interpreter(recorded):
  gcc = shutil.which('gcc')
  if gcc is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
    pytest.skip('no C toolchain / HIP headers on this box')
  libdir = os.path.join(ROOT, 'automl_amd')
  exe = os.path.join(recorded['dir'], 'edet_host')
  cmd = [gcc, '-std=c99', '-O1', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
         os.path.join(ROOT, 'tests', 'c_host', 'edet_host.c'), '-o', exe, '-L' + libdir, '-ledet_hip', '-L/opt/rocm/lib',
         '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
  r = subprocess.run(cmd, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  for mode in ('eager', 'graph'):
    out = os.path.join(recorded['dir'], 'out_' + mode)
    os.makedirs(out, exist_ok=True)
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    r = subprocess.run([exe, recorded['path'], out, mode, '1'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'ok' in r.stdout
    exp = recorded['expected']
    for k in ('params', 'ema', 'velocity', 'bn_state', 'loss_sums'):
      got = np.fromfile(os.path.join(out, k + '.step0.bin'), dtype=np.uint32)
      assert np.array_equal(got, exp[k].view(np.uint32).reshape(-1)), (mode, k)
    for k, dt in DET_DTYPES.items():
      got = np.fromfile(os.path.join(out, 'detections.%s.bin' % k), dtype=np.uint32)
      assert np.array_equal(got, exp['detections.' + k].view(np.uint32).reshape(-1)), (mode, k)
    c = recorded['config']
    for level in range(c.min_level, c.max_level + 1):
      for kind, ch in (('cls', c.num_classes * 9), ('box', 36)):
        name = '%s_outputs_%d' % (kind, level)
        got = np.fromfile(os.path.join(out, name + '.bin'), dtype=np.uint8)
        want = exp[name]
        assert got.shape == want.shape
        ld = (ch + 7) // 8 * 8
        eb = want.size // (2 * (SIZE >> level) ** 2 * ld)      # bytes per element: batch 2, (SIZE >> level)^2 pixels
        g = got.reshape(-1, ld * eb)[:, :ch * eb]
        w = want.reshape(-1, ld * eb)[:, :ch * eb]
        assert np.array_equal(g, w), (mode, name)
