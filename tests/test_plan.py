"""Step plans and the network-level C ABI (include/edet_net.h), the parts that need no GPU: the generated call stubs are
current, the library exports the network-level symbols, edet_anchors equals tf2/anchors.py's grid (through
automl_amd.anchors, itself bit-exact against the executed reference: tests/test_reference_kats.py), and the plan writer /
reader agree on the file format."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from automl_amd import _lib, anchors, hparams_config, net_c, plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_call_stubs_are_generated_from_the_current_header():
  rc = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'gen_plan_stubs.py'), '--check']).returncode
  assert rc == 0, 'automl_amd/csrc/plan_stubs.inc is stale: run scripts/gen_plan_stubs.py'


def test_every_operator_entry_point_has_a_stub_and_a_stream_position():
  stubs = open(os.path.join(ROOT, 'automl_amd', 'csrc', 'plan_stubs.inc')).read()
  pos = plan.stream_arg_index()
  for name, argtypes in _lib.SIGNATURES.items():
    assert 'return %s(' % name in stubs, name
    p = pos[name]
    if p is not None:
      assert argtypes[p] is ctypes.c_void_p, (name, p)


def test_library_exports_the_network_level_abi():
  src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'edet_net.h')).read(), flags=re.S)
  declared = set(re.findall(r'\b(edet_\w+)\s*\(', src)) - {'edet_allreduce_fn'}
  assert {'edet_create', 'edet_destroy', 'edet_forward', 'edet_train_step', 'edet_anchors', 'edet_dp_init'} <= declared
  assert declared - {'edet_net_buffer_name'} == set(net_c.NET_SIGNATURES), declared ^ set(net_c.NET_SIGNATURES)
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for name in declared:
    assert hasattr(lib, name), 'libedet_hip.so does not export %s' % name


@pytest.mark.parametrize('model,size', [('efficientdet-d0', 512), ('efficientdet-d0', 640), ('efficientdet-d1', 640),
                                        ('efficientdet-d7x', 1536), ('efficientdet-d0', (300, 200)),
                                        ('efficientdet-d2', (768, 1280))])
def test_edet_anchors_equals_the_python_anchor_grid(model, size):
  c = hparams_config.get_efficientdet_config(model)
  want = anchors.Anchors(c.min_level, c.max_level, c.num_scales, c.aspect_ratios, c.anchor_scale, size).boxes
  got = net_c.anchors(c.min_level, c.max_level, c.num_scales, c.aspect_ratios, c.anchor_scale, size)
  assert got.shape == want.shape
  assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_edet_anchors_other_scales_and_capacity_error():
  want = anchors.Anchors(2, 5, 2, [0.7, 1.4], 3.0, 96).boxes
  got = net_c.anchors(2, 5, 2, [0.7, 1.4], 3.0, 96)
  assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
  lib = net_c._lib_net()
  ar = (ctypes.c_double * 1)(1.0)
  count = ctypes.c_int64()
  buf = np.empty((4, 4), np.float32)
  rc = lib.edet_anchors(3, 7, 3, ar, 1, 4.0, 64, 64, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 4, ctypes.byref(count))
  assert rc != 0 and b'capacity' in lib.edet_last_error()


class _FakeRecorder(plan.Recorder):
  BLOCKS = [(0x10000, 4096), (0x20000, 1 << 20), (0x400000, 256)]

  def live_blocks(self):
    return list(self.BLOCKS)


def test_plan_writer_and_reader_agree(tmp_path):
  rec = _FakeRecorder()
  rec.begin('forward', main_stream=0)
  tv = _lib.TView(0x20000 + 512, 0x10000, 0x10000 + 64, None, 1, 2, 8, 8, 16, 16)
  nparts = ctypes.c_int(5)
  rec.on_call('edet_pw_fwd', (ctypes.byref(tv), 0x20000 + 4096, 16, None, 0x20000 + 8192, 24, 24, 0x400000,
                              ctypes.byref(nparts), 1, 0))
  ev = rec.event_record(0)
  rec.stream_wait(0x77, ev)
  rec.on_call('edet_zero', (0x10000 + 128, 256, 0x77))
  arr = (ctypes.c_void_p * 3)(0x20000, None, 0x400000 + 8)
  marr = (ctypes.c_int * 3)(1, 2, 0)
  rec.on_call('edet_bn_eval', (16, 0x10000, 0x10000, 1e-3, 0x10000, 0x10000, 0x10000, 0x10000, 0))
  rec.end()
  assert _lib.recorder is None
  blob, relocs = plan._blob_of(arr)
  assert len(blob) == 24 and relocs == [(0, 0x20000), (16, 0x400000 + 8)]
  assert plan._blob_of(marr) == (bytes(marr), [])
  rec.names['weights'] = (0x20000 + 4096, 1024)
  rec.props['batch'] = 2
  rec.dev_relocs.append((0x10000 + 8, 0x20000 + 4096))
  rec._initial = {0x10000: np.arange(4096, dtype=np.uint8)}
  path = str(tmp_path / 'fake.plan')
  summary = rec.write(path)
  assert summary['programs'] == {'forward': 5} and summary['streams'] == 2 and summary['events'] == 1
  got = plan.read_plan(path)
  assert got['entry_points'] == ['edet_pw_fwd', 'edet_zero', 'edet_bn_eval']
  ops = got['ops']['forward']
  assert [o[0] for o in ops] == ['call', 'evrec', 'wait', 'call', 'call']
  call0 = ops[0]
  assert call0[1] == 'edet_pw_fwd'
  kinds = [a[0] for a in call0[2]]
  assert kinds == ['b', 'p', 'i', 'n', 'p', 'i', 'i', 'p', 'b', 'i', 's']
  # buffers are numbered in order of first use: the 1 MiB block (tview.data) first, then the 4 KiB one (scale)
  tv_raw, tv_rel = call0[2][0][1], call0[2][0][2]
  assert len(tv_raw) == ctypes.sizeof(_lib.TView)
  assert tv_rel == [(0, 0, 512), (8, 1, 0), (16, 1, 64)]
  assert call0[2][1] == ('p', 0, 4096) and call0[2][7] == ('p', 2, 0) and call0[2][10] == ('s', 0)
  assert ops[1] == ('evrec', 0, 0) and ops[2] == ('wait', 1, 0)
  assert ops[3][2] == [('p', 1, 128), ('i', 256), ('s', 1)]
  assert ops[4][2][3] == ('f', 1e-3)
  assert got['names']['weights'] == (0, 4096, 1024) and got['names']['batch'][0] == plan.NULL_BUF and got['names']['batch'][1] == 2
  assert got['device_relocation_table'] == [(1, 8, 0, 4096)]
  sizes = [b[0] for b in got['buffers']]
  assert sizes == [1 << 20, 4096, 256]
  assert got['buffers'][0][1] == 0 and got['buffers'][1][1] % 256 == 0 and got['buffers'][1][1] > 0
  raw = open(path, 'rb').read()
  at = got['buffers'][1][1]
  assert raw[at:at + 4096] == bytes(np.arange(4096, dtype=np.uint8))


def test_pointer_outside_every_allocation_is_refused(tmp_path):
  rec = _FakeRecorder()
  rec.begin('forward', main_stream=0)
  rec.on_call('edet_zero', (0x999999, 16, 0))
  rec.end()
  with pytest.raises(_lib.EdetError, match='not inside a live allocation'):
    rec.write(str(tmp_path / 'bad.plan'))


def test_edet_create_refuses_a_file_that_is_not_a_plan(tmp_path):
  lib = net_c._lib_net()
  p = tmp_path / 'junk.plan'
  p.write_bytes(b'not a plan at all')
  h = ctypes.c_void_p()
  assert lib.edet_create(str(p).encode(), ctypes.byref(h)) != 0
  assert b'not a plan file' in lib.edet_last_error()
  assert lib.edet_create(str(tmp_path / 'missing.plan').encode(), ctypes.byref(h)) != 0
  assert b'cannot open' in lib.edet_last_error()
