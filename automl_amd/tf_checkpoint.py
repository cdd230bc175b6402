"""Reader / writer of TensorFlow checkpoints (TensorBundle V2: ``<prefix>.index`` + ``<prefix>.data-SSSSS-of-NNNNN``) without TensorFlow.

Checkpoint interchange with the reference: ``efficientdet/tf2/util_keras.py:108-203`` reads
its checkpoints through ``tf.train.list_variables`` / ``tf.train.load_checkpoint`` / ``tf.train.load_variable`` /
``tf.train.latest_checkpoint``; the functions of the same names below do that on the file format itself, and
``write_checkpoint`` produces files the reference's calls can read back.  TensorFlow is an un-vendored dependency of the
reference (``efficientdet/requirements.txt:8``, tensorflow>=2.10,<2.16), so the format is restated here from its published
definition:

  * ``.index`` is an immutable sorted string table in the LevelDB table format (tensorflow/core/lib/io/table*:
    prefix-compressed key/value blocks with restart points, each followed by a 1-byte compression tag and a masked
    CRC-32C, a metaindex block, an index block, and a 48-byte footer ending in the magic 0xdb4775248b80fb57).  Blocks may be
    snappy compressed (tag 1); the writer here stores them raw (tag 0), which every reader accepts.
  * key "" -> BundleHeaderProto {num_shards=1, endianness=2, version=3}; key <tensor name> -> BundleEntryProto
    {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32, masked CRC-32C of the tensor bytes)}
    (tensorflow/core/protobuf/tensor_bundle.proto).
  * the data shards hold the tensors' little-endian bytes back to back in key order; a DT_STRING tensor is stored as
    the varint64 lengths of its elements, a masked CRC-32C of those length bytes (4 bytes), then the strings.
  * object-based (TF2 ``tf.train.Checkpoint`` / Keras ``save_weights``) checkpoints carry the scalar string tensor
    ``_CHECKPOINTABLE_OBJECT_GRAPH``: a TrackableObjectGraph proto whose nodes list, per saved variable, its
    ``full_name`` (the variable's graph name) and ``checkpoint_key`` (tensorflow/core/protobuf/trackable_object_graph.proto).

Parity status: no checkpoint file exists under the reference tree and TensorFlow cannot be installed here, so this module
is pinned by round trips, by hand-assembled files that follow the format documents (tests/test_checkpoint.py, incl. a
snappy-compressed, multi-block, two-shard index) and by CRC-32C known answers (RFC 3720) -- not against a TF-written file.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
HEADER_KEY = ''

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 4, 5, 6, 7, 9, 10
DT_BFLOAT16, DT_UINT16, DT_HALF, DT_UINT32, DT_UINT64 = 14, 17, 19, 22, 23
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT16: np.int16,
             DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_, DT_UINT16: np.uint16, DT_HALF: np.float16,
             DT_UINT32: np.uint32, DT_UINT64: np.uint64}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items()}


# ------------------------------------------------------------------------------------------------ CRC-32C (Castagnoli)
def _make_crc_tables():
  poly = 0x82f63b78
  t0 = np.zeros(256, np.uint32)
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ poly if c & 1 else c >> 1
    t0[i] = c
  return t0


_CRC_T0_NP = _make_crc_tables()
_CRC_T0 = [int(x) for x in _CRC_T0_NP]
_CRC_CHUNK = 1024
_crc_shift_tables = None


def _shift_tables():
  """Four 256-entry tables of the linear map "register after _CRC_CHUNK zero bytes", one per register byte."""
  global _crc_shift_tables
  if _crc_shift_tables is None:
    st = np.concatenate([np.arange(256, dtype=np.uint32) << np.uint32(8 * k) for k in range(4)])
    for _ in range(_CRC_CHUNK):
      st = _CRC_T0_NP[st & 0xff] ^ (st >> np.uint32(8))
    _crc_shift_tables = [[int(x) for x in st[256 * k:256 * (k + 1)]] for k in range(4)]
  return _crc_shift_tables


def crc32c(data, crc=0):
  """CRC-32C (Castagnoli, reflected polynomial 0x82f63b78) of bytes-like ``data`` continuing from ``crc``
  (tensorflow/core/lib/hash/crc32c.h: Extend).  The register update is linear over GF(2), so long inputs are cut into
  1024-byte chunks whose registers advance in lockstep (numpy, one table lookup per byte position for all chunks at
  once) and are then chained through the precomputed "1024 zero bytes" map."""
  buf = np.frombuffer(memoryview(data).cast('B'), dtype=np.uint8)
  n = buf.size
  c = (crc ^ 0xffffffff) & 0xffffffff
  t0 = _CRC_T0
  nchunks = n // _CRC_CHUNK if n >= 16 * _CRC_CHUNK else 0
  if nchunks:
    cols = np.ascontiguousarray(buf[:nchunks * _CRC_CHUNK].reshape(nchunks, _CRC_CHUNK).T)
    st = np.zeros(nchunks, np.uint32)
    eight = np.uint32(8)
    for j in range(_CRC_CHUNK):
      st = _CRC_T0_NP[(st ^ cols[j]) & 0xff] ^ (st >> eight)
    s0, s1, s2, s3 = _shift_tables()
    for v in st.tolist():
      c = s0[c & 0xff] ^ s1[(c >> 8) & 0xff] ^ s2[(c >> 16) & 0xff] ^ s3[c >> 24] ^ v
  for b in buf[nchunks * _CRC_CHUNK:].tobytes():
    c = t0[(c ^ b) & 0xff] ^ (c >> 8)
  return c ^ 0xffffffff


def mask_crc(crc):
  """crc32c::Mask: rotate right by 15 bits and add a constant (stored CRCs of data that itself embeds CRCs)."""
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def unmask_crc(masked):
  rot = (masked - 0xa282ead8) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------ varints / protobuf wire
def _put_varint(out, v):
  v &= (1 << 64) - 1
  while v >= 0x80:
    out.append((v & 0x7f) | 0x80)
    v >>= 7
  out.append(v)


def _get_varint(buf, pos):
  shift = result = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7
    if shift > 63:
      raise ValueError('malformed varint')


def _parse_proto(buf):
  """Flat protobuf wire parse: list of (field number, wire type, value); value is int for varint / fixed, bytes for
  length-delimited."""
  out = []
  pos, n = 0, len(buf)
  while pos < n:
    tag, pos = _get_varint(buf, pos)
    field, wt = tag >> 3, tag & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wt == 2:
      ln, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + ln])
      if len(v) != ln:
        raise ValueError('truncated protobuf field')
      pos += ln
    elif wt == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wt)
    out.append((field, wt, v))
  return out


def _signed64(v):
  return v - (1 << 64) if v >= (1 << 63) else v


def _emit(out, field, wt, value):
  _put_varint(out, (field << 3) | wt)
  if wt == 0:
    _put_varint(out, value)
  elif wt == 2:
    _put_varint(out, len(value))
    out.extend(value)
  elif wt == 5:
    out.extend(struct.pack('<I', value))
  else:
    raise ValueError(wt)


def _encode_shape(shape):
  out = bytearray()
  for d in shape:
    dim = bytearray()
    _emit(dim, 1, 0, int(d))
    _emit(out, 2, 2, bytes(dim))
  return bytes(out)


def _decode_shape(buf):
  dims = []
  for field, _, v in _parse_proto(buf):
    if field == 2:
      size = 0
      for f2, _, v2 in _parse_proto(v):
        if f2 == 1:
          size = _signed64(v2)
      dims.append(size)
  return tuple(dims)


class BundleEntry(object):
  """BundleEntryProto of one tensor."""

  def __init__(self, dtype, shape, shard_id, offset, size, crc):
    self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c = dtype, shape, shard_id, offset, size, crc

  def encode(self):
    out = bytearray()
    _emit(out, 1, 0, self.dtype)
    _emit(out, 2, 2, _encode_shape(self.shape))
    if self.shard_id:
      _emit(out, 3, 0, self.shard_id)
    if self.offset:
      _emit(out, 4, 0, self.offset)
    _emit(out, 5, 0, self.size)
    _emit(out, 6, 5, self.crc32c)
    return bytes(out)

  @staticmethod
  def decode(buf):
    e = BundleEntry(0, (), 0, 0, 0, 0)
    for field, _, v in _parse_proto(buf):
      if field == 1:
        e.dtype = v
      elif field == 2:
        e.shape = _decode_shape(v)
      elif field == 3:
        e.shard_id = v
      elif field == 4:
        e.offset = v
      elif field == 5:
        e.size = v
      elif field == 6:
        e.crc32c = v
      elif field == 7:
        raise NotImplementedError('partitioned (sliced) variables are not supported')
    return e


# ------------------------------------------------------------------------------------------------ snappy (raw format)
def snappy_uncompress(data):
  """Raw snappy block decompression (format description: google/snappy format_description.txt)."""
  n, pos = _get_varint(data, 0)
  out = bytearray()
  ln = len(data)
  while pos < ln:
    tag = data[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:                                  # literal
      size = tag >> 2
      if size >= 60:
        nb = size - 59
        size = int.from_bytes(data[pos:pos + nb], 'little')
        pos += nb
      size += 1
      out += data[pos:pos + size]
      pos += size
      continue
    if kind == 1:
      length = 4 + ((tag >> 2) & 7)
      offset = ((tag >> 5) << 8) | data[pos]
      pos += 1
    elif kind == 2:
      length = (tag >> 2) + 1
      offset = data[pos] | (data[pos + 1] << 8)
      pos += 2
    else:
      length = (tag >> 2) + 1
      offset = int.from_bytes(data[pos:pos + 4], 'little')
      pos += 4
    if offset == 0 or offset > len(out):
      raise ValueError('corrupt snappy data')
    start = len(out) - offset
    for i in range(length):                        # may overlap its own output
      out.append(out[start + i])
  if len(out) != n:
    raise ValueError('corrupt snappy data: length %d, header says %d' % (len(out), n))
  return bytes(out)


# ------------------------------------------------------------------------------------------------ table reader
def _read_block(buf, offset, size, verify=True):
  raw = buf[offset:offset + size]
  if len(raw) != size or offset + size + 5 > len(buf):
    raise ValueError('truncated table block')
  ctype = buf[offset + size]
  stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
  if verify and unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
    raise ValueError('table block checksum mismatch')
  if ctype == 0:
    return bytes(raw)
  if ctype == 1:
    return snappy_uncompress(bytes(raw))
  raise ValueError('unknown block compression type %d' % ctype)


def _block_entries(block):
  """(key, value) pairs of one table block."""
  if len(block) < 4:
    raise ValueError('bad table block')
  num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * num_restarts
  if limit < 0:
    raise ValueError('bad table block restart array')
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_table(path, verify=True):
  """All (key bytes, value bytes) of a LevelDB-format table file, in key order."""
  with open(path, 'rb') as f:
    buf = f.read()
  if len(buf) < 48:
    raise ValueError('%s is too short to be a table file' % path)
  footer = buf[-48:]
  if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
    raise ValueError('%s: not a table file (bad magic number)' % path)
  _, p = _get_varint(footer, 0)           # metaindex handle: offset
  _, p = _get_varint(footer, p)           #                   size
  ioff, p = _get_varint(footer, p)
  isize, p = _get_varint(footer, p)
  out = []
  for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
    boff, q = _get_varint(handle, 0)
    bsize, q = _get_varint(handle, q)
    out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
  return out


# ------------------------------------------------------------------------------------------------ table writer
class _BlockBuilder(object):
  def __init__(self, restart_interval=16):
    self.buf = bytearray()
    self.restarts = [0]
    self.count = 0
    self.last = b''
    self.interval = restart_interval

  def add(self, key, value):
    shared = 0
    if self.count < self.interval:
      m = min(len(key), len(self.last))
      while shared < m and key[shared] == self.last[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.count = 0
    _put_varint(self.buf, shared)
    _put_varint(self.buf, len(key) - shared)
    _put_varint(self.buf, len(value))
    self.buf += key[shared:]
    self.buf += value
    self.last = key
    self.count += 1

  def finish(self):
    out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
    return out

  def size(self):
    return len(self.buf) + 4 * len(self.restarts) + 4


def write_table(path, items, block_size=65536):
  """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
  out = bytearray()

  def emit(block):
    off = len(out)
    out.extend(block)
    out.append(0)                                                    # kNoCompression
    out.extend(struct.pack('<I', mask_crc(crc32c(bytes(block) + b'\x00'))))
    handle = bytearray()
    _put_varint(handle, off)
    _put_varint(handle, len(block))
    return bytes(handle)

  index = _BlockBuilder(restart_interval=1)
  data = _BlockBuilder()
  prev = None
  for key, value in items:
    if prev is not None and not key > prev:
      raise ValueError('table keys must be strictly increasing')
    data.add(key, value)
    prev = key
    if data.size() >= block_size:
      index.add(prev, emit(data.finish()))
      data = _BlockBuilder()
  if data.count or prev is None:
    index.add(prev if prev is not None else b'', emit(data.finish()))
  meta_handle = emit(_BlockBuilder().finish())
  index_handle = emit(index.finish())
  footer = meta_handle + index_handle
  footer += b'\x00' * (40 - len(footer))
  footer += struct.pack('<Q', TABLE_MAGIC)
  out.extend(footer)
  with open(path, 'wb') as f:
    f.write(out)


# ------------------------------------------------------------------------------------------------ bundle
def _shard_name(prefix, shard, num_shards):
  return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def _encode_header(num_shards):
  out = bytearray()
  _emit(out, 1, 0, num_shards)
  # endianness = LITTLE (0) and the proto3 default are omitted; version {producer = 1 (kTensorBundleVersion)}
  ver = bytearray()
  _emit(ver, 1, 0, 1)
  _emit(out, 3, 2, bytes(ver))
  return bytes(out)


def _string_length_crc(lengths):
  """Running CRC-32C over the element lengths as tensor_bundle.cc computes it (WriteStringTensor / ReadStringTensor):
  every length enters as a FIXED-WIDTH little-endian integer -- 4 bytes when it fits a uint32, 8 bytes otherwise --
  not as the varint bytes that are stored."""
  crc = 0
  for ln in lengths:
    crc = crc32c(struct.pack('<I' if ln <= 0xffffffff else '<Q', ln), crc)
  return crc


def _encode_string_tensor(strings):
  """-> (bytes as stored in the data shard, unmasked CRC-32C of the BundleEntry).  Layout: [varint64 length] * n,
  4-byte masked checksum of the lengths, the string bytes.  The entry's CRC continues the length CRC over the 4 stored
  checksum bytes and then over every string (tensor_bundle.cc WriteStringTensor) -- it is NOT the CRC of the raw bytes
  on disk, because the lengths enter it in fixed width."""
  lengths = bytearray()
  for s in strings:
    _put_varint(lengths, len(s))
  crc = _string_length_crc([len(s) for s in strings])
  cks = struct.pack('<I', mask_crc(crc))
  crc = crc32c(cks, crc)
  for s in strings:
    crc = crc32c(s, crc)
  return bytes(lengths) + cks + b''.join(strings), crc


def _decode_string_tensor(raw, count, verify=True, entry_crc=None):
  """entry_crc: the BundleEntry's stored (masked) crc32c, checked when verify is set (ReadStringTensor)."""
  pos, lens = 0, []
  for _ in range(count):
    v, pos = _get_varint(raw, pos)
    lens.append(v)
  crc = _string_length_crc(lens)
  stored = raw[pos:pos + 4]
  if verify and unmask_crc(struct.unpack_from('<I', raw, pos)[0]) != crc:
    raise ValueError('string tensor: length checksum mismatch')
  crc = crc32c(bytes(stored), crc)
  pos += 4
  out = []
  for ln in lens:
    out.append(bytes(raw[pos:pos + ln]))
    crc = crc32c(out[-1], crc)
    pos += ln
  if verify and entry_crc is not None and unmask_crc(entry_crc) != crc:
    raise ValueError('string tensor: checksum mismatch')
  return out


def _bf16_to_f32(raw_u16):
  return (raw_u16.astype(np.uint32) << 16).view(np.float32)


class CheckpointReader(object):
  """``tf.train.load_checkpoint(prefix)``: has_tensor / get_tensor / get_variable_to_shape_map / ..dtype_map."""

  def __init__(self, prefix, verify=True):
    if not os.path.exists(prefix + '.index'):
      raise FileNotFoundError('no checkpoint index at %s.index' % prefix)
    self.prefix, self.verify = prefix, verify
    self.entries = {}
    self.num_shards = 1
    for key, value in read_table(prefix + '.index', verify):
      name = key.decode('utf-8')
      if name == HEADER_KEY:
        for field, _, v in _parse_proto(value):
          if field == 1:
            self.num_shards = v
          elif field == 2 and v != 0:
            raise NotImplementedError('big-endian checkpoints are not supported')
        continue
      self.entries[name] = BundleEntry.decode(value)
    self._shards = {}

  def has_tensor(self, name):
    return name in self.entries

  def get_variable_to_shape_map(self):
    return {k: list(e.shape) for k, e in self.entries.items()}

  def get_variable_to_dtype_map(self):
    return {k: e.dtype for k, e in self.entries.items()}

  def _shard(self, shard_id):
    m = self._shards.get(shard_id)
    if m is None:
      m = np.memmap(_shard_name(self.prefix, shard_id, self.num_shards), dtype=np.uint8, mode='r')
      self._shards[shard_id] = m
    return m

  def get_tensor(self, name):
    e = self.entries.get(name)
    if e is None:
      raise KeyError('Key %s not found in checkpoint %s' % (name, self.prefix))
    raw = self._shard(e.shard_id)[e.offset:e.offset + e.size]
    if len(raw) != e.size:
      raise ValueError('%s: data shard is truncated' % name)
    count = int(np.prod(e.shape)) if e.shape else 1
    if e.dtype == DT_STRING:
      strings = _decode_string_tensor(bytes(raw), count, self.verify, e.crc32c)
      if not e.shape:
        return strings[0]
      return np.array(strings, dtype=object).reshape(e.shape)
    if self.verify and unmask_crc(e.crc32c) != crc32c(raw):
      raise ValueError('%s: tensor checksum mismatch' % name)
    if e.dtype == DT_BFLOAT16:
      return _bf16_to_f32(np.frombuffer(raw, dtype='<u2', count=count)).reshape(e.shape)
    if e.dtype not in _NP_OF_DT:
      raise NotImplementedError('%s: dtype enum %d is not supported' % (name, e.dtype))
    dt = np.dtype(_NP_OF_DT[e.dtype]).newbyteorder('<')
    return np.frombuffer(raw, dtype=dt, count=count).reshape(e.shape).copy()


def _text_proto_unescape(text):
  """A quoted string of a text-format protobuf (the `checkpoint` state file is a CheckpointState in text format) ->
  str: C escapes act on BYTES (protobuf's text_format writes non-ASCII UTF-8 bytes as 3-digit octal escapes), the
  result is decoded as UTF-8; literal non-ASCII characters pass through."""
  out = bytearray()
  i, n = 0, len(text)
  simple = {'n': 10, 't': 9, 'r': 13, 'a': 7, 'b': 8, 'f': 12, 'v': 11, '\\': 92, '"': 34, "'": 39, '?': 63}
  while i < n:
    ch = text[i]
    if ch != '\\' or i + 1 >= n:
      out += ch.encode('utf-8')
      i += 1
      continue
    e = text[i + 1]
    if e in simple:
      out.append(simple[e])
      i += 2
    elif e in '01234567':
      j = i + 1
      while j < n and j < i + 4 and text[j] in '01234567':
        j += 1
      out.append(int(text[i + 1:j], 8) & 0xff)
      i = j
    elif e in 'xX':
      j = i + 2
      while j < n and j < i + 4 and text[j] in '0123456789abcdefABCDEF':
        j += 1
      out.append(int(text[i + 2:j] or '0', 16))
      i = j
    else:
      out += e.encode('utf-8')
      i += 2
  return out.decode('utf-8')


def load_checkpoint(ckpt_dir_or_file):
  """tf.train.load_checkpoint: a directory means its latest checkpoint."""
  path = ckpt_dir_or_file
  if os.path.isdir(path):
    latest = latest_checkpoint(path)
    if latest is None:
      raise FileNotFoundError("Couldn't find 'checkpoint' file or checkpoints in given directory %s" % path)
    path = latest
  return CheckpointReader(path)


def list_variables(ckpt_dir_or_file):
  """tf.train.list_variables: [(name, shape)] sorted by name."""
  r = load_checkpoint(ckpt_dir_or_file)
  return [(k, list(r.entries[k].shape)) for k in sorted(r.entries)]


def load_variable(ckpt_dir_or_file, name):
  """tf.train.load_variable (a trailing ':0' is dropped, as TF does)."""
  if name.endswith(':0'):
    name = name[:-2]
  return load_checkpoint(ckpt_dir_or_file).get_tensor(name)


def latest_checkpoint(checkpoint_dir, latest_filename=None):
  """tf.train.latest_checkpoint: the ``model_checkpoint_path`` of the directory's CheckpointState text proto
  (file ``checkpoint``), resolved against the directory; None when absent."""
  state = os.path.join(checkpoint_dir, latest_filename or 'checkpoint')
  if not os.path.exists(state):
    return None
  with open(state) as f:
    m = re.search(r'^\s*model_checkpoint_path:\s*"((?:[^"\\]|\\.)*)"', f.read(), re.M)
  if not m:
    return None
  path = _text_proto_unescape(m.group(1))
  if not os.path.isabs(path):
    path = os.path.join(checkpoint_dir, path)
  return path if os.path.exists(path + '.index') else None


def update_checkpoint_state(checkpoint_dir, prefix):
  """Writes the directory's ``checkpoint`` file naming ``prefix`` as the latest one (what tf.train.Saver /
  CheckpointManager leave behind and tf.train.latest_checkpoint reads)."""
  rel = os.path.relpath(prefix, checkpoint_dir)
  with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
    f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (rel, rel))


def write_checkpoint(prefix, tensors):
  """tensors: name -> numpy array (or bytes / str for a scalar DT_STRING tensor).  One data shard; tensors in key
  order, as BundleWriter lays them out."""
  names = sorted(tensors, key=lambda s: s.encode('utf-8'))
  items = [(b'', _encode_header(1))]
  offset = 0
  os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
  with open(_shard_name(prefix, 0, 1), 'wb') as f:
    for name in names:
      if name == HEADER_KEY:
        raise ValueError('the empty tensor name is reserved for the bundle header')
      v = tensors[name]
      if isinstance(v, (bytes, str)):
        raw, crc = _encode_string_tensor([v.encode('utf-8') if isinstance(v, str) else v])
        entry = BundleEntry(DT_STRING, (), 0, offset, len(raw), mask_crc(crc))
      else:
        a = np.asarray(v, order='C')          # (ascontiguousarray would turn a scalar into shape [1])
        if a.dtype.byteorder == '>':
          a = a.astype(a.dtype.newbyteorder('<'))
        dt = _DT_OF_NP.get(np.dtype(a.dtype.name))
        if dt is None:
          raise TypeError('%s: dtype %s cannot be stored' % (name, a.dtype))
        raw = a.tobytes()
        entry = BundleEntry(dt, tuple(a.shape), 0, offset, len(raw), mask_crc(crc32c(raw)))
      f.write(raw)
      offset += len(raw)
      items.append((name.encode('utf-8'), entry.encode()))
  write_table(prefix + '.index', items)
  return prefix


# ------------------------------------------------------------------------------------------------ object graph
def parse_object_graph(serialized):
  """TrackableObjectGraph -> list of nodes: {'children': [(local_name, node_id)], 'attributes': [(name, full_name,
  checkpoint_key)], 'slot_variables': [(original_variable_node_id, slot_name, slot_variable_node_id)]}."""
  nodes = []
  for field, _, v in _parse_proto(serialized):
    if field != 1:
      continue
    node = {'children': [], 'attributes': [], 'slot_variables': []}
    for f2, _, v2 in _parse_proto(v):
      if f2 == 1:
        node_id, local = 0, ''
        for f3, _, v3 in _parse_proto(v2):
          if f3 == 1:
            node_id = v3
          elif f3 == 2:
            local = v3.decode('utf-8')
        node['children'].append((local, node_id))
      elif f2 == 2:
        name = full = key = ''
        for f3, _, v3 in _parse_proto(v2):
          if f3 == 1:
            name = v3.decode('utf-8')
          elif f3 == 2:
            full = v3.decode('utf-8')
          elif f3 == 3:
            key = v3.decode('utf-8')
        node['attributes'].append((name, full, key))
      elif f2 == 3:
        orig = slot_node = 0
        slot = ''
        for f3, _, v3 in _parse_proto(v2):
          if f3 == 1:
            orig = v3
          elif f3 == 2:
            slot = v3.decode('utf-8')
          elif f3 == 3:
            slot_node = v3
        node['slot_variables'].append((orig, slot, slot_node))
    nodes.append(node)
  return nodes


def encode_object_graph(nodes):
  """Inverse of parse_object_graph."""
  out = bytearray()
  for node in nodes:
    nb = bytearray()
    for local, node_id in node.get('children', []):
      cb = bytearray()
      if node_id:
        _emit(cb, 1, 0, node_id)
      _emit(cb, 2, 2, local.encode('utf-8'))
      _emit(nb, 1, 2, bytes(cb))
    for name, full, key in node.get('attributes', []):
      ab = bytearray()
      _emit(ab, 1, 2, name.encode('utf-8'))
      _emit(ab, 2, 2, full.encode('utf-8'))
      _emit(ab, 3, 2, key.encode('utf-8'))
      _emit(nb, 2, 2, bytes(ab))
    for orig, slot, slot_node in node.get('slot_variables', []):
      sb = bytearray()
      if orig:
        _emit(sb, 1, 0, orig)
      _emit(sb, 2, 2, slot.encode('utf-8'))
      if slot_node:
        _emit(sb, 3, 0, slot_node)
      _emit(nb, 3, 2, bytes(sb))
    _emit(out, 1, 2, bytes(nb))
  return bytes(out)


def object_graph_variables(reader):
  """For an object-based checkpoint: ({variable full_name: checkpoint_key}, {(variable full_name, slot name):
  checkpoint_key of the slot variable}, {checkpoint_key: top-level attribute the variable hangs under})."""
  nodes = parse_object_graph(reader.get_tensor(OBJECT_GRAPH_KEY))
  by_name, node_var = {}, {}
  for i, node in enumerate(nodes):
    for name, full, key in node['attributes']:
      if name == 'VARIABLE_VALUE':
        node_var[i] = (full, key)
        if full and full not in by_name:
          by_name[full] = key
  slots = {}
  for node in nodes:
    for orig, slot, slot_node in node['slot_variables']:
      if orig in node_var and slot_node in node_var:
        slots[(node_var[orig][0], slot)] = node_var[slot_node][1]
  top = {key: key.split('/')[0] for _, key in node_var.values()}
  return by_name, slots, top
