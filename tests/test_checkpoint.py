"""Checkpoint interchange: the TensorBundle reader / writer (automl_amd/tf_checkpoint.py) and the
mirror of the reference's restore logic (automl_amd/util_keras.py vs efficientdet/tf2/util_keras.py:67-203).

No checkpoint file ships with the reference and TensorFlow cannot be installed here, so the FORMAT is pinned by the
published CRC-32C known answers, by files assembled by hand from the format's definition (snappy-compressed blocks,
several data blocks, two data shards -- none of which the writer under test produces) and by round trips; the RESTORE
SEMANTICS are pinned statement by statement against util_keras.restore_ckpt (insertion order of the EMA keys, the
reference's error messages, skip_mismatch, exclude_layers, the hub fallback)."""
import os
import struct

import numpy as np
import pytest

from automl_amd import hparams_config, netspec, tf_checkpoint as tfc, util_keras


# ------------------------------------------------------------------ format
def test_crc32c_known_answers():
  """RFC 3720 appendix B.4 vectors and the 'check' value of the CRC-32C catalogue entry."""
  assert tfc.crc32c(b'\x00' * 32) == 0x8a9136aa
  assert tfc.crc32c(b'\xff' * 32) == 0x62a8ab43
  assert tfc.crc32c(bytes(range(32))) == 0x46dd794e
  assert tfc.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
  assert tfc.crc32c(b'123456789') == 0xe3069283
  # leveldb / TensorFlow crc32c_test: Mask / Unmask
  crc = tfc.crc32c(b'foo')
  assert tfc.mask_crc(crc) != crc and tfc.unmask_crc(tfc.mask_crc(crc)) == crc
  assert tfc.unmask_crc(tfc.unmask_crc(tfc.mask_crc(tfc.mask_crc(crc)))) == crc


def test_crc32c_chunked_path_equals_bytewise_and_extends():
  data = np.random.default_rng(0).integers(0, 256, 200001, dtype=np.uint8).tobytes()
  c = 0xffffffff
  for b in data:
    c = tfc._CRC_T0[(c ^ b) & 0xff] ^ (c >> 8)
  assert tfc.crc32c(data) == c ^ 0xffffffff
  assert tfc.crc32c(data[70000:], tfc.crc32c(data[:70000])) == tfc.crc32c(data)


def _snappy_literals_and_copies(raw):
  """A valid snappy stream for `raw` built by hand: 40-byte literals, and a 2-byte-offset copy wherever the next 8 bytes
  repeat the 8 bytes before them (the index blocks below have such runs)."""
  out = bytearray()
  tfc._put_varint(out, len(raw))
  pos = 0
  while pos < len(raw):
    if pos >= 8 and raw[pos:pos + 8] == raw[pos - 8:pos] and pos + 8 <= len(raw):
      out.append(((8 - 1) << 2) | 2)            # copy, length 8, 2-byte offset
      out += struct.pack('<H', 8)
      pos += 8
      continue
    n = min(40, len(raw) - pos)
    out.append((n - 1) << 2)                    # literal
    out += raw[pos:pos + n]
    pos += n
  return bytes(out)


def test_snappy_uncompress():
  raw = b'abcdefgh' * 5 + b'tail' + bytes(range(200))
  assert tfc.snappy_uncompress(_snappy_literals_and_copies(raw)) == raw
  # overlapping copy (run-length): literal 'ab' then copy offset 2 length 10 -> 'ab' * 6
  stream = bytes([12, (2 - 1) << 2]) + b'ab' + bytes([((10 - 4) << 2) | 1, 2])
  assert tfc.snappy_uncompress(stream) == b'ab' * 6
  # long literal with a 1-byte length
  raw = bytes(range(256)) * 1
  stream = bytearray()
  tfc._put_varint(stream, len(raw))
  stream += bytes([60 << 2, len(raw) - 1]) + raw[:256]
  assert tfc.snappy_uncompress(bytes(stream)) == raw
  with pytest.raises(ValueError):
    tfc.snappy_uncompress(bytes([4, 0 << 2]) + b'a')          # header says 4 bytes, stream holds 1


def _hand_table(path, blocks, compress):
  """Writes a table file from explicit blocks [[(key, value), ...], ...] with this test's own encoder: no prefix
  sharing except inside a block's second entry onwards, restart interval 2, optional snappy."""
  out = bytearray()

  def emit(entries, interval):
    buf, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
      shared = 0
      if i % interval == 0:
        restarts.append(len(buf))
      else:
        while shared < min(len(k), len(last)) and k[shared] == last[shared]:
          shared += 1
      tfc._put_varint(buf, shared)
      tfc._put_varint(buf, len(k) - shared)
      tfc._put_varint(buf, len(v))
      buf += k[shared:] + v
      last = k
    if not restarts:
      restarts = [0]
    buf += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    body, tag = bytes(buf), 0
    if compress:
      body, tag = _snappy_literals_and_copies(bytes(buf)), 1
    off = len(out)
    out.extend(body)
    out.append(tag)
    out.extend(struct.pack('<I', tfc.mask_crc(tfc.crc32c(body + bytes([tag])))))
    h = bytearray()
    tfc._put_varint(h, off)
    tfc._put_varint(h, len(body))
    return bytes(h)
  index = [(blk[-1][0] + b'\x00', emit(blk, 2)) for blk in blocks]     # separator > last key of the block
  meta = emit([], 1)
  idx = emit(index, 1)
  footer = meta + idx
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', tfc.TABLE_MAGIC)
  out.extend(footer)
  with open(path, 'wb') as f:
    f.write(out)


@pytest.mark.parametrize('compress', [False, True], ids=['raw', 'snappy'])
def test_hand_assembled_two_shard_bundle(tmp_path, compress):
  """An index the writer under test would never produce: three data blocks (prefix-compressed keys, restart interval
  2), optionally snappy, two data shards, bfloat16 / int64 / string tensors."""
  prefix = str(tmp_path / 'model.ckpt-7')
  rng = np.random.default_rng(1)
  t = {'a/kernel': rng.standard_normal((3, 3, 2, 4)).astype(np.float32),
       'a/kernel/ExponentialMovingAverage': rng.standard_normal((3, 3, 2, 4)).astype(np.float32),
       'b/beta': rng.standard_normal((5,)).astype(np.float32),
       'b/steps': np.asarray(12345678901, np.int64),
       'c/half': rng.standard_normal((4,)).astype(np.float32)}
  bf = (t['c/half'].view(np.uint32) >> 16).astype('<u2')            # stored as DT_BFLOAT16
  shard_of = {'a/kernel': 0, 'a/kernel/ExponentialMovingAverage': 1, 'b/beta': 1, 'b/steps': 0, 'c/half': 1}
  data = [bytearray(), bytearray()]
  entries = []
  for name in sorted(t):
    raw = bf.tobytes() if name == 'c/half' else t[name].tobytes()
    dt = tfc.DT_BFLOAT16 if name == 'c/half' else tfc._DT_OF_NP[t[name].dtype]
    sh = shard_of[name]
    e = tfc.BundleEntry(dt, t[name].shape, sh, len(data[sh]), len(raw), tfc.mask_crc(tfc.crc32c(raw)))
    data[sh] += raw
    entries.append((name.encode(), e.encode()))
  sraw, scrc = tfc._encode_string_tensor([b'hello', b'', b'bundle'])
  entries.append((b'd/strings', tfc.BundleEntry(tfc.DT_STRING, (3,), 0, len(data[0]), len(sraw),
                                                tfc.mask_crc(scrc)).encode()))
  data[0] += sraw
  header = bytearray()
  tfc._emit(header, 1, 0, 2)                                         # num_shards = 2
  for i in range(2):
    with open(tfc._shard_name(prefix, i, 2), 'wb') as f:
      f.write(data[i])
  items = [(b'', bytes(header))] + entries
  _hand_table(prefix + '.index', [items[:2], items[2:5], items[5:]], compress)

  r = tfc.CheckpointReader(prefix)
  assert r.num_shards == 2
  assert sorted(r.get_variable_to_shape_map()) == sorted(list(t) + ['d/strings'])
  for name in t:
    got = r.get_tensor(name)
    want = t[name] if name != 'c/half' else (bf.astype(np.uint32) << 16).view(np.float32)
    assert got.shape == want.shape and np.array_equal(got, want), name
  assert list(r.get_tensor('d/strings')) == [b'hello', b'', b'bundle']
  assert tfc.list_variables(prefix)[0] == ('a/kernel', [3, 3, 2, 4])
  assert tfc.load_variable(prefix, 'b/steps:0') == 12345678901
  with pytest.raises(KeyError):
    r.get_tensor('nope')
  # a flipped data byte is caught by the entry's CRC
  with open(tfc._shard_name(prefix, 1, 2), 'r+b') as f:
    f.seek(3)
    b = f.read(1)
    f.seek(3)
    f.write(bytes([b[0] ^ 0x40]))
  with pytest.raises(ValueError, match='checksum'):
    tfc.CheckpointReader(prefix).get_tensor('a/kernel/ExponentialMovingAverage')


def _crc32c_bitwise(data, crc=0):
  """Bit-at-a-time CRC-32C (reflected polynomial 0x82f63b78), written here independently of tf_checkpoint's tables."""
  c = crc ^ 0xffffffff
  for b in bytes(data):
    c ^= b
    for _ in range(8):
      c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
  return c ^ 0xffffffff


def test_string_tensor_checksums_follow_tensor_bundle_cc(tmp_path):
  """DT_STRING layout of tensorflow/core/util/tensor_bundle/tensor_bundle.cc (WriteStringTensor / ReadStringTensor),
  assembled here byte by byte from that definition with an independent CRC: varint64 lengths, then Mask(crc) of the
  lengths taken as FIXED-WIDTH little-endian uint32 (not of the varint bytes), then the strings; the BundleEntry's
  crc32c continues the length CRC over the 4 stored checksum bytes and over every string, then is masked.  A 300-byte
  string makes the varint (2 bytes) and the fixed-width form (4 bytes) differ, which is what the first version of the
  encoder got wrong.  No TF-written file exists here: this pins the code against the format definition."""
  strings = [b'x' * 300, b'', b'object graph']
  varints = b'\xac\x02' + b'\x00' + b'\x0c'
  crc = 0
  for s_ in strings:
    crc = _crc32c_bitwise(len(s_).to_bytes(4, 'little'), crc)
  masked = (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff
  want_raw = varints + masked.to_bytes(4, 'little') + b''.join(strings)
  entry_crc = _crc32c_bitwise(masked.to_bytes(4, 'little'), crc)
  for s_ in strings:
    entry_crc = _crc32c_bitwise(s_, entry_crc)
  raw, got_crc = tfc._encode_string_tensor(strings)
  assert raw == want_raw and got_crc == entry_crc
  assert _crc32c_bitwise(varints) != crc            # the varint-byte checksum is a different number
  assert tfc._decode_string_tensor(raw, 3, True, tfc.mask_crc(entry_crc)) == strings
  with pytest.raises(ValueError):
    tfc._decode_string_tensor(raw, 3, True, tfc.mask_crc(entry_crc ^ 1))
  bad = bytearray(raw)
  bad[4] ^= 0xff                                      # a checksum byte
  with pytest.raises(ValueError):
    tfc._decode_string_tensor(bytes(bad), 3, True, None)
  # through the bundle writer / reader: a scalar string tensor (what _CHECKPOINTABLE_OBJECT_GRAPH is)
  prefix = str(tmp_path / 'ckpt-1')
  tfc.write_checkpoint(prefix, {'graph': b'g' * 70000, 'w': np.arange(3, dtype=np.float32)})
  r = tfc.CheckpointReader(prefix)
  assert r.get_tensor('graph') == b'g' * 70000
  e = r.entries['graph']
  c2 = _crc32c_bitwise((70000).to_bytes(4, 'little'))
  m2 = (((c2 >> 15) | (c2 << 17)) + 0xa282ead8) & 0xffffffff
  assert tfc.unmask_crc(e.crc32c) == _crc32c_bitwise(b'g' * 70000, _crc32c_bitwise(m2.to_bytes(4, 'little'), c2))


def test_checkpoint_state_path_unescape(tmp_path):
  """The `checkpoint` state file is a text-format CheckpointState: C escapes on bytes, UTF-8 underneath."""
  d = tmp_path / 'mod\u00e8le'
  d.mkdir()
  prefix = str(d / 'ckpt-3')
  tfc.write_checkpoint(prefix, {'w': np.zeros(2, np.float32)})
  for line in ('model_checkpoint_path: "ckpt-3"', 'model_checkpoint_path: "%s"' % prefix,
               'model_checkpoint_path: "%s"' % ''.join(chr(b) if b < 128 else '\\%03o' % b for b in prefix.encode())):
    (d / 'checkpoint').write_text(line + '\n', encoding='utf-8')
    assert tfc.latest_checkpoint(str(d)) == prefix, line
  assert tfc._text_proto_unescape(r'a\"b\\c\x41\101') == 'a"b\\cAA'


def test_write_read_round_trip_many_blocks(tmp_path):
  """2000 variables: the writer cuts several index blocks; every tensor comes back bit for bit; the files have the
  layout constants of the format (footer magic, header entry first, offsets ascending in key order)."""
  prefix = str(tmp_path / 'sub' / 'ckpt')
  rng = np.random.default_rng(2)
  t = {'scope_%03d/layer_%d/kernel' % (i // 7, i): rng.standard_normal((i % 5 + 1, 3)).astype(np.float32)
       for i in range(2000)}
  t['global_step'] = np.asarray(77, np.int64)
  t[tfc.OBJECT_GRAPH_KEY] = b'\x0a\x00'
  tfc.write_checkpoint(prefix, t)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
  table = tfc.read_table(prefix + '.index')
  assert table[0][0] == b'' and [k for k, _ in table] == sorted(k for k, _ in table) and len(table) == 2003
  offs = [tfc.BundleEntry.decode(v).offset for _, v in table[1:]]
  assert offs == sorted(offs) and offs[0] == 0
  r = tfc.load_checkpoint(prefix)
  for k, v in t.items():
    got = r.get_tensor(k)
    assert (got == v) if isinstance(v, bytes) else np.array_equal(got, v), k
  assert os.path.getsize(tfc._shard_name(prefix, 0, 1)) == offs[-1] + tfc.BundleEntry.decode(table[-1][1]).size


def test_latest_checkpoint(tmp_path):
  d = str(tmp_path)
  assert tfc.latest_checkpoint(d) is None
  tfc.write_checkpoint(os.path.join(d, 'ckpt-3'), {'v': np.zeros(2, np.float32)})
  tfc.update_checkpoint_state(d, os.path.join(d, 'ckpt-3'))
  assert tfc.latest_checkpoint(d) == os.path.join(d, 'ckpt-3')
  assert tfc.list_variables(d) == [('v', [2])]
  with open(os.path.join(d, 'checkpoint'), 'w') as f:
    f.write('model_checkpoint_path: "gone-9"\n')
  assert tfc.latest_checkpoint(d) is None


# ------------------------------------------------------------------ restore semantics
class _Net(object):
  """Stands for EfficientDetNet (no optimizer): the variable inventory of a real d0 NetSpec, values kept on the host."""

  def __init__(self):
    self.config = hparams_config.get_efficientdet_config('efficientdet-d0')
    self.spec = netspec.NetSpec(self.config)
    self.values, self.ema = {}, {}

  def set_weights(self, values):
    self.values.update({k: np.array(v) for k, v in values.items()})

  def get_weights(self):
    return dict(self.values)


class _TrainNet(_Net):
  def train_step(self, data):
    raise NotImplementedError

  def set_ema_weights(self, values):
    self.ema.update({k: np.array(v) for k, v in values.items()})

  def get_ema_weights(self):
    return {**self.values, **self.ema}


def _d0_values(seed):
  spec = netspec.NetSpec(hparams_config.get_efficientdet_config('efficientdet-d0'))
  rng = np.random.default_rng(seed)
  return {p.name: rng.standard_normal(p.shape).astype(np.float32) for p in spec.params}, spec


def test_restore_name_based_checkpoint_with_ema(tmp_path):
  vals, spec = _d0_values(3)
  shadows = {k: v + 1 for k, v in vals.items()}
  t = dict(vals)
  t.update({util_keras.average_name(k): v for k, v in shadows.items()})
  t['global_step'] = np.asarray(5, np.int64)
  prefix = tfc.write_checkpoint(str(tmp_path / 'model.ckpt'), t)

  # no optimizer, ema on: the shadow overwrites the variable (util_keras.py:165-180, opt_ema_fn = lambda var: var)
  net = _Net()
  util_keras.restore_ckpt(net, prefix, ema_decay=0.9998, skip_mismatch=False)
  assert set(net.values) == set(vals)
  assert all(np.array_equal(net.values[k], shadows[k]) for k in vals)
  # ema off: the plain values
  net = _Net()
  util_keras.restore_ckpt(net, prefix, ema_decay=0, skip_mismatch=False)
  assert all(np.array_equal(net.values[k], vals[k]) for k in vals)
  # training model: variables plain, shadows of the trainable ones into the optimizer's average slots
  net = _TrainNet()
  util_keras.restore_ckpt(net, prefix, ema_decay=0.9998, skip_mismatch=False)
  assert all(np.array_equal(net.values[k], vals[k]) for k in vals)
  trainable = {p.name for p in spec.params if p.trainable}
  assert set(net.ema) == trainable and all(np.array_equal(net.ema[k], shadows[k]) for k in trainable)
  # '_' loads nothing; a directory resolves through its state file
  net = _Net()
  util_keras.restore_ckpt(net, '_')
  assert not net.values
  tfc.update_checkpoint_state(str(tmp_path), prefix)
  util_keras.restore_ckpt(net, str(tmp_path), ema_decay=0)
  assert len(net.values) == len(vals)


def test_restore_name_based_mismatch_handling(tmp_path):
  vals, _ = _d0_values(4)
  missing = 'box_net/box-predict/bias'
  reshaped = 'class_net/class-predict/pointwise_kernel'       # a 90-class head loaded into another class count
  t = {k: v for k, v in vals.items() if k != missing}
  t[reshaped] = np.zeros((1, 1, 64, 9 * 20), np.float32)
  prefix = tfc.write_checkpoint(str(tmp_path / 'm'), t)
  net = _Net()
  util_keras.restore_ckpt(net, prefix, ema_decay=0, skip_mismatch=True)
  assert missing not in net.values and reshaped not in net.values and len(net.values) == len(vals) - 2
  with pytest.raises(ValueError, match=r'Shape mismatch: %s, expected \(1, 1, 64, 810\), but got \(1, 1, 64, 180\)' % reshaped):
    util_keras.restore_ckpt(_Net(), prefix, ema_decay=0, skip_mismatch=False)
  del t[reshaped]
  t[reshaped] = vals[reshaped]
  prefix = tfc.write_checkpoint(str(tmp_path / 'm2'), t)
  with pytest.raises(KeyError, match='Not found %s in' % missing):
    util_keras.restore_ckpt(_Net(), prefix, ema_decay=0, skip_mismatch=False)
  # with ema on and skip_mismatch off, a checkpoint without shadows is an error (the reference looks every key up)
  with pytest.raises(KeyError, match='ExponentialMovingAverage'):
    util_keras.restore_ckpt(_Net(), tfc.write_checkpoint(str(tmp_path / 'm3'), vals), ema_decay=0.9, skip_mismatch=False)


def _object_checkpoint(path, vals, spec, attr_of, with_slots):
  """An object-based checkpoint as tf.train.Checkpoint(model) lays it out: node 0 = root, one node per variable (with
  its full_name and key '<attr>/v<i>/.ATTRIBUTES/VARIABLE_VALUE'), optimizer 'average' slots as slot variables."""
  nodes = [{'children': [], 'attributes': [], 'slot_variables': []}]
  tensors = {}
  for i, p in enumerate(spec.params):
    key = '%s/v%d/.ATTRIBUTES/VARIABLE_VALUE' % (attr_of(p.name), i)
    nodes.append({'attributes': [('VARIABLE_VALUE', p.name, key)]})
    nodes[0]['children'].append(('v%d' % i, len(nodes) - 1))
    tensors[key] = vals[p.name]
  if with_slots:
    opt = {'children': [], 'attributes': [], 'slot_variables': []}
    nodes.append(opt)
    opt_id = len(nodes) - 1
    nodes[0]['children'].append(('optimizer', opt_id))
    for i, p in enumerate(spec.params):
      if not p.trainable:
        continue
      key = '%s/v%d/.OPTIMIZER_SLOT/optimizer/average/.ATTRIBUTES/VARIABLE_VALUE' % (attr_of(p.name), i)
      nodes.append({'attributes': [('VARIABLE_VALUE', p.name + '/average', key)]})
      opt['slot_variables'].append((i + 1, 'average', len(nodes) - 1))
      tensors[key] = vals[p.name] * 2
  tensors[tfc.OBJECT_GRAPH_KEY] = tfc.encode_object_graph(nodes)
  return tfc.write_checkpoint(path, tensors)


def _attr(name):
  for prefix, attr in (('class_net/', 'class_net'), ('box_net/', 'box_net'), ('fpn_cells/', 'fpn_cells'),
                       ('resample_p', 'resample_layers')):
    if name.startswith(prefix):
      return attr
  return 'backbone'


def test_restore_object_based_checkpoint(tmp_path):
  vals, spec = _d0_values(5)
  prefix = _object_checkpoint(str(tmp_path / 'ckpt-1'), vals, spec, _attr, with_slots=True)
  assert tfc.list_variables(prefix)[0][0] == tfc.OBJECT_GRAPH_KEY        # what util_keras.py:132-134 keys on
  nodes = tfc.parse_object_graph(tfc.load_checkpoint(prefix).get_tensor(tfc.OBJECT_GRAPH_KEY))
  assert len(nodes) == 2 + len(spec.params) + sum(p.trainable for p in spec.params)
  net = _TrainNet()
  util_keras.restore_ckpt(net, prefix)
  assert all(np.array_equal(net.values[k], vals[k]) for k in vals)
  assert all(np.array_equal(net.ema[p.name], vals[p.name] * 2) for p in spec.params if p.trainable)
  # exclude_layers: the class head keeps its own values (fine-tuning on another label set, tf2/train.py)
  net = _Net()
  util_keras.restore_ckpt(net, prefix, exclude_layers=['class_net'])
  assert net.values and not any(k.startswith('class_net/') for k in net.values)
  assert all(k in net.values for k in vals if not k.startswith('class_net/'))
  # a shape the model does not have is an error in this branch (tf.train.Checkpoint.restore raises too)
  bad = dict(vals)
  bad['box_net/box-predict/bias'] = np.zeros(7, np.float32)
  with pytest.raises(ValueError, match='Shape mismatch'):
    util_keras.restore_ckpt(_Net(), _object_checkpoint(str(tmp_path / 'ckpt-2'), bad, spec, _attr, False))


def test_hub_checkpoint_fallback(tmp_path):
  """An EfficientDetNetTrainHub checkpoint matches no attribute of the model; the reference then reads the keys
  HUB_CPT_NAME spells out (util_keras.py:24-26,83-105,152-157)."""
  vals, spec = _d0_values(6)
  tensors = {}
  for p in spec.params:
    if p.name.startswith('class_net/class-predict/'):
      key = 'classes/' + p.name[len('class_net/class-predict/'):].replace('/', '.S')
    elif p.name.startswith('box_net/box-predict/'):
      key = 'boxes/' + p.name[len('box_net/box-predict/'):].replace('/', '.S')
    else:
      key = 'base_model/' + (p.name + ':0').replace('/', '.S')
    tensors[key + '/.ATTRIBUTES/VARIABLE_VALUE'] = vals[p.name]
  tensors[tfc.OBJECT_GRAPH_KEY] = tfc.encode_object_graph([{'children': [('base_model', 1)]}, {}])
  prefix = tfc.write_checkpoint(str(tmp_path / 'hub'), tensors)
  net = _Net()
  util_keras.restore_ckpt(net, prefix)
  assert all(np.array_equal(net.values[k], vals[k]) for k in vals)


def test_save_ckpt_layout_and_round_trip(tmp_path):
  vals, spec = _d0_values(7)
  net = _TrainNet()
  net.set_weights(vals)
  net.set_ema_weights({p.name: vals[p.name] * 0.5 for p in spec.params if p.trainable})
  prefix = util_keras.save_ckpt(net, str(tmp_path / 'out' / 'model.ckpt-100'), global_step=100)
  names = dict(tfc.list_variables(str(tmp_path / 'out')))
  assert names['global_step'] == [] and len(names) == 2 * len(vals) + 1
  assert names['efficientnet-b0/stem/conv2d/kernel/ExponentialMovingAverage'] == [3, 3, 3, 32]
  back = _TrainNet()
  util_keras.restore_ckpt(back, prefix, skip_mismatch=False)
  assert all(np.array_equal(back.values[k], vals[k]) for k in vals)
  assert all(np.array_equal(back.ema[p.name], vals[p.name] * 0.5) for p in spec.params if p.trainable)


# ------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_train_save_restore_round_trip_on_device(tmp_path):
  """Two optimizer steps on the GPU, save_ckpt, restore_ckpt into a fresh training model and into a fresh inference
  model: variables, BatchNorm statistics and EMA shadows come back bit for bit; the inference model restored with
  ema_decay > 0 computes with the shadows, exactly what the reference's eval / export path does (util_keras.py:165-180)."""
  import torch
  from automl_amd import efficientdet_net, train_lib
  from tests.test_gpu_network import make_labels, perturbed_params
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('moving_average_decay=0.9')
  vals = perturbed_params(config, 11)
  rng = np.random.default_rng(12)
  images = rng.standard_normal((2, 128, 128, 3)).astype(np.float32)
  labels = make_labels(config, 2, 128, 13)
  net = train_lib.EfficientDetNetTrain(config=config, dtype='f32', params=vals)
  for _ in range(2):
    net.train_step((images, labels))
  prefix = util_keras.save_ckpt(net, str(tmp_path / 'model.ckpt-2'), global_step=2)
  w, ema = net.get_weights(), net.get_ema_weights()
  trainable = [n for n, _, tr in util_keras.model_variables(net) if tr]
  assert any(not np.array_equal(w[n], ema[n]) for n in trainable)

  back = train_lib.EfficientDetNetTrain(config=config, dtype='f32')
  util_keras.restore_ckpt(back, str(tmp_path), skip_mismatch=False)
  back(torch.from_numpy(images), training=False)               # builds the executor with the restored values
  w2, ema2 = back.get_weights(), back.get_ema_weights()
  assert all(np.array_equal(w[n], w2[n]) for n in w)
  assert all(np.array_equal(ema[n], ema2[n]) for n in trainable)

  infer = efficientdet_net.EfficientDetNet(config=config, dtype='f32')
  util_keras.restore_ckpt(infer, prefix, ema_decay=0.9998, skip_mismatch=False)
  cls_e, _ = infer(torch.from_numpy(images), training=False)
  want = efficientdet_net.EfficientDetNet(config=config, dtype='f32', params=ema)
  cls_w, _ = want(torch.from_numpy(images), training=False)
  # (two executions of the same network agree to fp32 rounding, not bit for bit: the SE pooled sums are atomics)
  def rel(a, b):
    return float((a.float() - b.float()).abs().max()) / float(b.float().abs().max())
  assert max(rel(a, b) for a, b in zip(cls_e, cls_w)) <= 1e-5, [rel(a, b) for a, b in zip(cls_e, cls_w)]
  assert all(np.array_equal(infer.get_weights()[n], ema[n]) for n in ema)
  plain = efficientdet_net.EfficientDetNet(config=config, dtype='f32')
  util_keras.restore_ckpt(plain, prefix, ema_decay=0, skip_mismatch=False)
  plain(torch.from_numpy(images), training=False)
  assert all(np.array_equal(plain.get_weights()[n], w[n]) for n in w)
