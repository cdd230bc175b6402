"""Step plans: the launches of an engine pass, recorded at the C ABI and written to a file that a host WITHOUT a Python
interpreter replays (csrc/net_runtime.cpp behind include/edet_net.h: edet_create / edet_forward / edet_train_step).

Why a recorded plan and not a second engine in C++: every launch of a step already goes through `_lib.call` (the layers,
and since round 6 the clears, the side chain's gradient join and the loss normalizer: edet_zero / edet_axpy_clear /
edet_loss_normalizer), the arguments of a step are a pure function of (config, batch, image size, dtype), and the captured
hipGraph the Python host replays is exactly such a recording made by the HIP runtime.  The plan is the same recording made
one level up, where it can be written to disk: buffers (size, optional initial contents: variables, optimizer slots, moving
statistics, tables), named handles into them (images, targets, logits, hyper-parameters), and per program ('forward',
'train_step') the list of C-ABI calls with every device pointer expressed as (buffer, offset), the stream forks / joins of
the two head chains as event operations, and a marker where the data-parallel gradient exchange goes
(tf2/train_lib.py:675-683 -> edet_dp_init).

The reference interfaces this gives a compiled host: efficientdet_keras.EfficientDetNet.__init__/call
(efficientdet/tf2/efficientdet_keras.py:790-799, 893-915) and EfficientDetNetTrain.train_step (tf2/train_lib.py:606-684).

File layout (little endian), version 1:
  'EDETPLAN' u32 version u32 nbuf u32 nnames u32 nstreams u32 nevents u32 nprog u32 nfn u32 ndevreloc
  nfn   x { u16 len, name }                                   entry-point names (index = fn id)
  nbuf  x { u64 bytes, u64 init_offset (0 = none: zero-filled) }
  nnames x { u16 len, name, u32 buf, u64 offset, u64 bytes }   (buf 0xffffffff: an integer property, value in `offset`)
  ndevreloc x { u32 buf, u64 at, u32 target buf, u64 target offset }   device pointers stored INSIDE initial contents (the
                                                                       descriptor table of edet_cast_batch): patched after upload
  nprog x { u16 len, name, u32 nops, ops }
  op: u8 kind
    0 CALL   u16 fn, u8 nargs, args: u8 type { 0 int64 | 1 double | 2 devptr u32 buf u64 off (buf 0xffffffff = NULL)
                                                | 3 stream u32 idx | 4 host blob u32 nbytes, bytes, u16 nreloc x {u32 at, u32 buf, u64 off} | 5 NULL }
    1 EVENT_RECORD u32 event u32 stream      2 STREAM_WAIT u32 stream u32 event      3 ALLREDUCE_SUM_F32 u32 buf u64 off u64 count u32 stream
  initial contents, 256-byte aligned, at the offsets the buffer table names.
"""
import bisect
import ctypes
import os
import re
import struct

import numpy as np
import torch

from automl_amd import _lib

MAGIC = b'EDETPLAN'
VERSION = 1
NULL_BUF = 0xffffffff
_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'edet_hip.h')


def header_prototypes(path=_HEADER):
  """name -> list of parameter declarations, parsed from include/edet_hip.h (the same parse as tests/test_abi.py)."""
  src = re.sub(r'/\*.*?\*/', '', open(path).read(), flags=re.S)
  out = {}
  for m in re.finditer(r'\b(?:int|const char\*)\s+(edet_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
    out[m.group(1)] = [a.strip() for a in m.group(2).split(',') if a.strip() and a.strip() != 'void']
  return out


def stream_arg_index():
  """name -> position of the `void* stream` parameter (None if the entry point takes none)."""
  out = {}
  for name, args in header_prototypes().items():
    idx = [i for i, a in enumerate(args) if re.search(r'\bvoid\s*\*\s*stream$', a)]
    out[name] = idx[0] if idx else None
  return out


def _blob_of(obj):
  """(bytes, [(offset, pointer value)]) of a ctypes structure / array / scalar passed by reference."""
  raw = ctypes.string_at(ctypes.addressof(obj), ctypes.sizeof(obj))
  relocs = []
  if isinstance(obj, ctypes.Structure):
    for fname, ftype in obj._fields_:
      if ftype is ctypes.c_void_p:
        v = getattr(obj, fname)
        if v:
          relocs.append((getattr(type(obj), fname).offset, int(v)))
  elif isinstance(obj, ctypes.Array) and obj._type_ is ctypes.c_void_p:
    for i in range(len(obj)):
      if obj[i]:
        relocs.append((8 * i, int(obj[i])))
  elif isinstance(obj, ctypes.c_void_p):
    if obj.value:
      relocs.append((0, int(obj.value)))
  return raw, relocs


class Recorder(object):
  """Collects the operations of one or more programs; `install()` hooks `_lib.call` and the engine's fork / join."""

  def __init__(self):
    self.programs = []          # (name, ops)
    self._ops = None
    self.streams = {}           # stream handle -> index (0 = the caller's stream)
    self.nevents = 0
    self.names = {}             # name -> (pointer, bytes)
    self.props = {}             # name -> integer property of the network (batch, image size, levels, ...)
    self.dev_relocs = []        # (address of an 8-byte slot in device memory, device pointer stored there)
    self._stream_pos = stream_arg_index()
    self._keep = []             # tensors that must stay allocated until the plan is written

  # ---- recording ------------------------------------------------------------------------------------------------
  def begin(self, program, main_stream=None):
    assert self._ops is None, 'a program is already being recorded'
    main = torch.cuda.current_stream().cuda_stream if main_stream is None else main_stream
    if not self.streams:
      self.streams[main] = 0
    assert self.streams.get(main) == 0, 'every program must be recorded on the same main stream'
    self._ops = []
    self._program = program
    _lib.recorder = self

  def end(self):
    _lib.recorder = None
    self.programs.append((self._program, self._ops))
    self._ops = None

  def _stream(self, handle):
    handle = int(handle or 0)
    if handle not in self.streams:
      self.streams[handle] = len(self.streams)
    return self.streams[handle]

  def on_call(self, name, args):
    if self._ops is None:
      return
    argtypes = _lib.SIGNATURES[name]
    spos = self._stream_pos.get(name)
    enc = []
    for i, (a, t) in enumerate(zip(args, argtypes)):
      if i == spos:
        enc.append(('s', self._stream(a)))
      elif t in (_lib.c_int, _lib.c_int64, ctypes.c_size_t):
        enc.append(('i', int(a)))
      elif t in (_lib.c_float, _lib.c_double):
        enc.append(('f', float(a)))
      elif t is _lib.c_void_p and (isinstance(a, (ctypes.Array, ctypes.Structure)) or hasattr(a, '_obj')):
        # a HOST array / out-parameter behind a void* parameter (edet_preprocess_infer's mean, std and scale)
        enc.append(('b',) + _blob_of(getattr(a, '_obj', a)))
      elif t is _lib.c_void_p:
        if isinstance(a, ctypes.c_void_p):
          a = a.value
        enc.append(('p', int(a)) if a else ('n',))
      elif a is None:
        enc.append(('n',))
      else:      # pointer to a structure / array / scalar on the host
        obj = getattr(a, '_obj', a)
        enc.append(('b',) + _blob_of(obj))
    self._ops.append(('call', name, enc))

  def event_record(self, stream_handle):
    """-> event id; the event is recorded on the stream at this point of the program."""
    if self._ops is None:
      return None
    ev = self.nevents
    self.nevents += 1
    self._ops.append(('evrec', ev, self._stream(stream_handle)))
    return ev

  def stream_wait(self, stream_handle, ev):
    if self._ops is not None and ev is not None:
      self._ops.append(('wait', self._stream(stream_handle), ev))

  def allreduce(self, tensor, stream_handle):
    """Marks the data-parallel gradient exchange (SUM over the replicas, in place) at this point of the program."""
    if self._ops is not None:
      self._ops.append(('allreduce', tensor.data_ptr(), tensor.numel(), self._stream(stream_handle)))

  def name_buffer(self, name, tensor):
    self.names[name] = (tensor.data_ptr(), tensor.numel() * tensor.element_size())
    self._keep.append(tensor)

  def device_table(self, tensor, relocs):
    """A device buffer whose (initial) contents hold device pointers: relocs = [(byte offset, pointer value)]."""
    self._keep.append(tensor)
    for at, p in relocs:
      if p:
        self.dev_relocs.append((tensor.data_ptr() + at, int(p)))

  # ---- resolution against the allocator + writing ----------------------------------------------------------------
  def live_blocks(self):
    """Sorted [(address, bytes)] of the caching allocator's allocated blocks."""
    out = []
    for seg in torch.cuda.memory_snapshot():
      addr = seg['address']
      for b in seg['blocks']:
        a = b.get('address', addr)
        if b['state'] == 'active_allocated':
          out.append((a, b['size']))
        addr = a + b['size']
    out.sort()
    return out

  def snapshot_initial_state(self, persistent, max_bytes=32 << 20):
    """Copies to the host what the replay must start from: every live block that holds a `persistent` tensor, and every
    other live block of at most max_bytes (tables, scalars, inputs).  Call AFTER a warm-up pass (all buffers exist) and
    BEFORE the recorded passes."""
    torch.cuda.synchronize()
    blocks = self.live_blocks()
    starts = [a for a, _ in blocks]
    must = set()
    for t in persistent:
      i = bisect.bisect_right(starts, t.data_ptr()) - 1
      assert i >= 0 and t.data_ptr() < blocks[i][0] + blocks[i][1], 'persistent tensor outside the allocator'
      must.add(i)
    self._initial = {}
    for i, (a, n) in enumerate(blocks):
      if i in must or n <= max_bytes:
        from automl_amd import net_c
        self._initial[a] = net_c.copy_to_host(a, n)

  def write(self, path):
    """Resolves every recorded device pointer to (buffer, offset) and writes the plan."""
    assert self._ops is None
    blocks = self.live_blocks()
    starts = [a for a, _ in blocks]
    used = {}          # block index -> buffer id

    def resolve(p):
      i = bisect.bisect_right(starts, p) - 1
      if i < 0 or p >= blocks[i][0] + blocks[i][1]:
        raise _lib.EdetError('plan: device pointer 0x%x is not inside a live allocation (a temporary freed during '
                             'the recorded pass?)' % p)
      if i not in used:
        used[i] = len(used)
      return used[i], p - blocks[i][0]

    fn_ids = {}
    progs = []
    for pname, ops in self.programs:
      body = bytearray()
      for op in ops:
        if op[0] == 'call':
          _, name, enc = op
          fid = fn_ids.setdefault(name, len(fn_ids))
          body += struct.pack('<BHB', 0, fid, len(enc))
          for e in enc:
            if e[0] == 'i':
              body += struct.pack('<Bq', 0, e[1])
            elif e[0] == 'f':
              body += struct.pack('<Bd', 1, e[1])
            elif e[0] == 'p':
              b, off = resolve(e[1])
              body += struct.pack('<BIQ', 2, b, off)
            elif e[0] == 's':
              body += struct.pack('<BI', 3, e[1])
            elif e[0] == 'n':
              body += struct.pack('<B', 5)
            else:
              _, raw, relocs = e
              body += struct.pack('<BI', 4, len(raw)) + raw + struct.pack('<H', len(relocs))
              for at, p in relocs:
                b, off = resolve(p)
                body += struct.pack('<IIQ', at, b, off)
        elif op[0] == 'evrec':
          body += struct.pack('<BII', 1, op[1], op[2])
        elif op[0] == 'wait':
          body += struct.pack('<BII', 2, op[1], op[2])
        else:
          b, off = resolve(op[1])
          body += struct.pack('<BIQQI', 3, b, off, op[2], op[3])
      progs.append((pname, len(ops), bytes(body)))
    names = []
    for n, (p, nbytes) in sorted(self.names.items()):
      b, off = resolve(p)
      names.append((n, b, off, nbytes))
    for n, v in sorted(self.props.items()):
      names.append((n, NULL_BUF, int(v), 0))
    devrel = []
    for at, p in self.dev_relocs:
      devrel.append(resolve(at) + resolve(p))

    def s16(text):
      raw = text.encode()
      return struct.pack('<H', len(raw)) + raw

    by_id = sorted(used.items(), key=lambda kv: kv[1])
    head = bytearray(MAGIC + struct.pack('<IIIIIIII', VERSION, len(by_id), len(names), len(self.streams), self.nevents,
                                         len(progs), len(fn_ids), len(devrel)))
    for name, _ in sorted(fn_ids.items(), key=lambda kv: kv[1]):
      head += s16(name)
    table_at = len(head)
    head += b'\0' * (16 * len(by_id))
    for n, b, off, nbytes in names:
      head += s16(n) + struct.pack('<IQQ', b, off, nbytes)
    for b, at, tb, toff in devrel:
      head += struct.pack('<IQIQ', b, at, tb, toff)
    for pname, nops, body in progs:
      head += s16(pname) + struct.pack('<I', nops) + body
    pos = (len(head) + 255) // 256 * 256
    table = bytearray()
    payload = []
    for bi, _ in by_id:
      a, n = blocks[bi]
      init = getattr(self, '_initial', {}).get(a)
      if init is not None and len(init) == n:
        table += struct.pack('<QQ', n, pos)
        payload.append((pos, init))
        pos = (pos + n + 255) // 256 * 256
      else:
        table += struct.pack('<QQ', n, 0)
    head[table_at:table_at + len(table)] = table
    with open(path, 'wb') as f:
      f.write(head)
      for at, init in payload:
        f.seek(at)
        f.write(init.tobytes())
    return {'buffers': len(by_id), 'initialised': len(payload), 'bytes': pos, 'programs': {p: n for p, n, _ in progs},
            'entry_points': len(fn_ids), 'streams': len(self.streams), 'events': self.nevents}


def read_summary(path):
  """Header of a plan file (CPU: the format test and `python -m automl_amd.plan FILE`)."""
  with open(path, 'rb') as f:
    raw = f.read(8 + 32)
    assert raw[:8] == MAGIC, 'not a plan file'
    version, nbuf, nnames, nstreams, nevents, nprog, nfn, ndevreloc = struct.unpack('<IIIIIIII', raw[8:])

    def s16():
      (n,) = struct.unpack('<H', f.read(2))
      return f.read(n).decode()
    fns = [s16() for _ in range(nfn)]
    bufs = [struct.unpack('<QQ', f.read(16)) for _ in range(nbuf)]
    names = {}
    for _ in range(nnames):
      n = s16()
      names[n] = struct.unpack('<IQQ', f.read(20))
  return {'version': version, 'entry_points': fns, 'buffers': bufs, 'names': names, 'streams': nstreams,
          'events': nevents, 'programs': nprog, 'device_relocations': ndevreloc}


def read_plan(path):
  """The whole plan decoded (CPU; the mirror of csrc/net_runtime.cpp's loader): summary + 'ops': program -> list of
  ('call', entry point, [args]) / ('evrec', event, stream) / ('wait', stream, event) / ('allreduce', buf, off, count, stream);
  args: ('i', v) ('f', v) ('p', buf, off) ('s', idx) ('b', bytes, [(at, buf, off)]) ('n',)."""
  out = read_summary(path)
  with open(path, 'rb') as f:
    data = f.read()
  pos = [8 + 32]

  def take(fmt):
    vals = struct.unpack_from('<' + fmt, data, pos[0])
    pos[0] += struct.calcsize('<' + fmt)
    return vals if len(vals) > 1 else vals[0]

  def s16():
    n = take('H')
    v = data[pos[0]:pos[0] + n].decode()
    pos[0] += n
    return v
  for _ in out['entry_points']:
    s16()
  pos[0] += 16 * len(out['buffers'])
  for _ in range(len(out['names'])):
    s16()
    pos[0] += 20
  devrel = [take('IQIQ') for _ in range(out['device_relocations'])]
  progs = {}
  for _ in range(out['programs']):
    name = s16()
    ops = []
    for _ in range(take('I')):
      kind = take('B')
      if kind == 0:
        fid, nargs = take('HB')
        args = []
        for _ in range(nargs):
          t = take('B')
          if t == 0:
            args.append(('i', take('q')))
          elif t == 1:
            args.append(('f', take('d')))
          elif t == 2:
            args.append(('p',) + take('IQ'))
          elif t == 3:
            args.append(('s', take('I')))
          elif t == 4:
            n = take('I')
            raw = data[pos[0]:pos[0] + n]
            pos[0] += n
            args.append(('b', raw, [take('IIQ') for _ in range(take('H'))]))
          else:
            assert t == 5, t
            args.append(('n',))
        ops.append(('call', out['entry_points'][fid], args))
      elif kind == 1:
        ops.append(('evrec',) + take('II'))
      elif kind == 2:
        ops.append(('wait',) + take('II'))
      else:
        assert kind == 3, kind
        ops.append(('allreduce',) + take('IQQI'))
    progs[name] = ops
  out['ops'] = progs
  out['device_relocation_table'] = devrel
  return out


def train_pass(eng, images, dlabels, learning_rate, ema_decay):
  """One training step in the structure the plan records (and edet_train_step replays): per-step scalars, forward with
  batch statistics, device-side normalizer, losses + backward, L2 + clip applied locally, [gradient exchange], update."""
  eng.set_hyper(learning_rate, ema_decay)
  eng.forward(images, training=True)
  eng.set_normalizer(dlabels['mean_num_positives'])
  glabels = dict(dlabels)
  glabels['normalizer'] = 'device'
  eng.loss_backward(glabels)
  eng.optimizer_local(True)
  if _lib.recorder is not None:
    _lib.recorder.allreduce(eng.grads_flat, eng.stream)
  eng.optimizer_apply(bool(ema_decay), True)


class _Detect(object):
  """Buffers and launches of the `detect` program: EfficientDetModel.call with pre_mode='infer', post_mode='global'
  (efficientdet_keras.py:920-1000) -- edet_preprocess_infer, the network, the level outputs without their padding columns,
  pre_nms (tf2/postprocess.py:120-157), global NMS + clip + rescale (:375-406).  Every buffer exists BEFORE the recorded
  pass (the Python host's own path, automl_amd/postprocess.py, allocates its temporaries inside the calls)."""

  def __init__(self, net, eng, images, raw_hw):
    from automl_amd import anchors as anchors_lib, postprocess, utils
    c = eng.config
    self.eng, self.params = eng, c.as_dict()
    dev = images.device
    b = int(images.shape[0])
    rh, rw = int(raw_hw[0]), int(raw_hw[1])
    self.b, self.rh, self.rw = b, rh, rw
    self.oh, self.ow = int(images.shape[1]), int(images.shape[2])
    self.raw = torch.zeros((b, rh, rw, 3), dtype=torch.uint8, device=dev)
    self.images = torch.empty_like(images)
    self.mean = (ctypes.c_float * 3)(*[float(v) for v in np.broadcast_to(np.asarray(c.mean_rgb, np.float32).reshape(-1), (3,))])
    self.std = (ctypes.c_float * 3)(*[float(v) for v in np.broadcast_to(np.asarray(c.stddev_rgb, np.float32).reshape(-1), (3,))])
    self.scale = ctypes.c_float(0.0)
    self.tdt = _lib.EDET_BF16 if images.dtype == torch.bfloat16 else _lib.EDET_F32
    a = anchors_lib.Anchors(c.min_level, c.max_level, c.num_scales, c.aspect_ratios, c.anchor_scale, (self.oh, self.ow))
    self.anchors = torch.as_tensor(np.asarray(a.boxes, np.float32)).to(dev).contiguous()
    self.n = int(self.anchors.shape[0])
    self.na = eng.spec.num_anchors
    self.k = int(c.nms_configs.get('max_nms_inputs', 0) or 0)
    kk = self.k if self.k > 0 else self.n
    self.boxes = torch.empty((b, kk, 4), dtype=torch.float32, device=dev)
    self.scores = torch.empty((b, kk), dtype=torch.float32, device=dev)
    self.classes = torch.empty((b, kk), dtype=torch.int32, device=dev)
    self.ws_topk = None
    if self.k > 0:
      need = ctypes.c_size_t(0)
      _lib.call('edet_pre_nms_topk_workspace_bytes', b, self.k, ctypes.byref(need))
      self.ws_topk = torch.empty((max(need.value, 8),), dtype=torch.uint8, device=dev)
    self.cfg = postprocess._tf_nms_cfg(c.nms_configs)
    self.m = int(self.cfg.max_output_size)
    need = ctypes.c_size_t(0)
    _lib.call('edet_nms_workspace_bytes', b, kk, 1, self.m, ctypes.byref(need))
    self.ws_nms = torch.empty((max(need.value, 8),), dtype=torch.uint8, device=dev)
    self.out_index = torch.empty((b, self.m), dtype=torch.int32, device=dev)
    self.out_score = torch.empty((b, self.m), dtype=torch.float32, device=dev)
    self.valid = torch.empty((b,), dtype=torch.int32, device=dev)
    self.nms_boxes = torch.empty((b, self.m, 4), dtype=torch.float32, device=dev)
    self.nms_scores = torch.empty((b, self.m), dtype=torch.float32, device=dev)
    self.nms_classes = torch.empty((b, self.m), dtype=torch.float32, device=dev)
    # image_scales of preprocess_infer: one value per image, a function of the two sizes only (the library returns it on
    # the host); kept on the device as postprocess_global's rescale wants it
    self.scales = torch.zeros((b,), dtype=torch.float32, device=dev)
    self.compact = None      # per level (cls, box) without padding columns: made by the first pass (needs the views)
    self.clip_hw = utils.parse_image_size((self.oh, self.ow))

  def persistent(self):
    return [self.raw, self.anchors, self.scales]

  def run(self):
    eng = self.eng
    st = eng.stream
    _lib.call('edet_preprocess_infer', self.raw.data_ptr(), 0, self.b, self.rh, self.rw, self.oh, self.ow, self.mean, self.std,
              self.images.data_ptr(), ctypes.byref(self.scale), self.tdt, st)
    if not float(self.scales[0]):
      self.scales.fill_(self.scale.value)      # (before the recorded pass: the warm-up pass does this once)
    eng.forward(self.images, training=False)
    views = list(zip(eng.cls_views, eng.box_views))
    if self.compact is None:
      self.compact = [tuple(torch.empty((v.raw.n, v.raw.h, v.raw.w, v.raw.c), dtype=v.raw.data.dtype, device=self.raw.device)
                            for v in pair) for pair in views]
    dts = {pair[i].raw.data.dtype for pair in views for i in (0, 1)}
    assert len(dts) == 1, 'class and box outputs of one storage type'
    dt = dts.pop()
    for pair, outs in zip(views, self.compact):
      for v, o in zip(pair, outs):
        r = v.raw
        _lib.call('edet_compact_rows', _lib.ptr(r.data), r.n * r.h * r.w, r.c, r.ld, _lib.ptr(o), r.data.element_size(), st)
    nlev = len(views)
    cp = (ctypes.c_void_p * nlev)(*[o[0].data_ptr() for o in self.compact])
    bp = (ctypes.c_void_p * nlev)(*[o[1].data_ptr() for o in self.compact])
    lp = (ctypes.c_int * nlev)(*[pair[0].raw.h * pair[0].raw.w for pair in views])
    edt = _lib.EDET_BF16 if dt == torch.bfloat16 else _lib.EDET_F32
    nc = eng.config.num_classes
    if self.k > 0:
      _lib.call('edet_pre_nms_topk', cp, bp, lp, nlev, self.b, self.na, nc, self.anchors.data_ptr(), edt, self.k,
                self.ws_topk.data_ptr(), self.ws_topk.numel(), self.boxes.data_ptr(), self.scores.data_ptr(),
                self.classes.data_ptr(), st)
    else:
      _lib.call('edet_pre_nms', cp, bp, lp, nlev, self.b, self.na, nc, self.anchors.data_ptr(), edt, self.boxes.data_ptr(),
                self.scores.data_ptr(), self.classes.data_ptr(), st)
    kk = int(self.scores.shape[1])
    _lib.call('edet_nms', self.boxes.data_ptr(), self.scores.data_ptr(), self.classes.data_ptr(), self.b, kk, 1,
              ctypes.byref(self.cfg), self.ws_nms.data_ptr(), self.ws_nms.numel(), self.out_index.data_ptr(),
              self.out_score.data_ptr(), self.valid.data_ptr(), st)
    _lib.call('edet_nms_gather', self.boxes.data_ptr(), self.classes.data_ptr(), self.out_index.data_ptr(),
              self.out_score.data_ptr(), self.b, kk, self.m, _lib.NMS_PAD_INDEX0, float(self.clip_hw[0]), float(self.clip_hw[1]),
              self.scales.data_ptr(), self.nms_boxes.data_ptr(), self.nms_scores.data_ptr(), self.nms_classes.data_ptr(), st)


def record_network(net, images, labels=None, path='efficientdet.plan', learning_rate=0.01, ema_decay=0.0,
                   max_init_bytes=32 << 20, detect_raw_hw=None, raw_images=None):
  """Records `forward` (EfficientDetNet.call, inference BatchNorm) and -- with labels -- `train_step` of a network
  (efficientdet_net.EfficientDetNet / train_lib.EfficientDetNetTrain) on device tensors `images` [B,H,W,3] and the label
  dictionary of train_step, and writes the plan.  Returns (summary, expected): `expected` holds what the Python host
  computed in the recorded passes (numpy), for a replay to be compared with.

  detect_raw_hw = (H, W): also records `detect` (EfficientDetModel.call on raw uint8 images of that size: preprocessing,
  network, global-NMS post-processing; edet_detect); raw_images [B,H,W,3] uint8 are the recorded inputs (default zeros).

  The state the plan starts from is the network's state after one un-recorded warm-up pass of each program.
  """
  b, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
  eng = net._ensure_engine(b, h, w)
  assert eng.sync_bn is None and eng._overlap_reduce is None, 'plans record the default step structure'
  images = net._to_device_images(images, eng)
  dl = None
  if labels is not None:
    dl = net._labels_to_device(labels, eng)
    assert 'mean_num_positives' in dl, "the recorded step computes the normalizer on the device: labels['mean_num_positives']"

  det = None
  if detect_raw_hw is not None:
    det = _Detect(net, eng, images, detect_raw_hw)
    if raw_images is not None:
      det.raw.copy_(torch.as_tensor(raw_images).to(det.raw.device))
  # warm-up: every buffer of both programs exists afterwards
  eng.forward(images, training=False)
  if det is not None:
    det.run()
  if dl is not None:
    train_pass(eng, images, dl, learning_rate, ema_decay)
  torch.cuda.synchronize()
  rec = Recorder()
  persistent = [eng.params_flat, eng.velocity, eng.ema, eng.state_flat, eng.seg_flags, eng.seg_offsets, eng.seg_factor,
                eng.hyper, images]
  if dl is not None:
    persistent += [t for t in dl.values() if torch.is_tensor(t)]
  if getattr(eng.arena, 'adam_v', None) is not None:
    persistent.append(eng.arena.adam_v)
  if det is not None:
    persistent += det.persistent()
  rec.snapshot_initial_state(persistent, max_init_bytes)
  expected = {}
  # every program re-makes the compute copies of the variables (and, in inference, the BatchNorm vectors): a replayed step
  # follows other replayed steps, whose updates the host-side version counter of this engine has not seen
  eng._cast_version = -1
  rec.begin('forward')
  eng.forward(images, training=False)
  rec.end()
  torch.cuda.synchronize()
  # the logits buffers as stored: [B, h, w, ld] with ld = channels rounded up to 8 (fp32 in an inference pass of a bf16
  # engine, see Engine.logits_f32); the first `channels` of a pixel are the outputs of efficientdet_keras.py:893-915
  for li, (cv, bv) in enumerate(zip(eng.cls_views, eng.box_views)):
    level = eng.config.min_level + li
    for kind, v in (('cls', cv), ('box', bv)):
      t = v.raw.data
      rec.name_buffer('%s_outputs_%d' % (kind, level), t)
      rec.props['%s_outputs_%d.channels' % (kind, level)] = v.raw.c
      rec.props['%s_outputs_%d.ld' % (kind, level)] = v.raw.ld
      rec.props['%s_outputs_%d.height' % (kind, level)] = v.raw.h
      rec.props['%s_outputs_%d.width' % (kind, level)] = v.raw.w
      rec.props['%s_outputs_%d.elem_bytes' % (kind, level)] = t.element_size()
      expected['%s_outputs_%d' % (kind, level)] = t.detach().view(torch.uint8).cpu().numpy().reshape(-1).copy()
  rec.props.update({'batch': b, 'height': h, 'width': w, 'min_level': eng.config.min_level,
                    'max_level': eng.config.max_level, 'num_classes': eng.config.num_classes,
                    'num_anchors': eng.spec.num_anchors, 'storage_elem_bytes': eng.esize,
                    'num_train_elems': int(eng.params_flat.numel())})
  if eng._cast_table is not None:
    names, dev, _, _ = eng._cast_table
    flat = [it for n in names for it in eng._cast_items[n]]
    rec.device_table(dev, [(32 * i + 8 * j, it[j]) for i, it in enumerate(flat) for j in (0, 1)])
  if det is not None:
    eng._cast_version = -1
    rec.begin('detect')
    det.run()
    rec.end()
    torch.cuda.synchronize()
    for name, t in (('raw_images', det.raw), ('detections.boxes', det.nms_boxes), ('detections.scores', det.nms_scores),
                    ('detections.classes', det.nms_classes), ('detections.valid_len', det.valid),
                    ('image_scales', det.scales)):
      rec.name_buffer(name, t)
    rec.props.update({'raw_height': det.rh, 'raw_width': det.rw, 'max_output_size': det.m})
    for k in ('boxes', 'scores', 'classes', 'valid_len'):
      t = {'boxes': det.nms_boxes, 'scores': det.nms_scores, 'classes': det.nms_classes, 'valid_len': det.valid}[k]
      expected['detections.' + k] = t.cpu().numpy().copy()
    rec._keep.append(det)
  if dl is not None:
    eng._cast_version = -1
    rec.begin('train_step')
    train_pass(eng, images, dl, learning_rate, ema_decay)
    rec.end()
    torch.cuda.synchronize()
    expected['params'] = eng.params_flat.cpu().numpy().copy()
    expected['ema'] = eng.ema.cpu().numpy().copy()
    expected['velocity'] = eng.velocity.cpu().numpy().copy()
    expected['bn_state'] = eng.state_flat.cpu().numpy().copy()
    expected['loss_sums'] = eng.loss_sums.cpu().numpy().copy()
    for k, t in dl.items():
      if torch.is_tensor(t):
        rec.name_buffer(k, t)
  rec.name_buffer('images', images)
  rec.name_buffer('params', eng.params_flat)
  rec.name_buffer('ema', eng.ema)
  rec.name_buffer('velocity', eng.velocity)
  rec.name_buffer('bn_state', eng.state_flat)
  rec.name_buffer('loss_sums', eng.loss_sums)
  rec.name_buffer('hyper', eng.hyper)
  for scope, (mask, _) in eng.drop_masks.items():
    rec.name_buffer('drop_mask:%s' % scope, mask)
  return rec.write(path), expected


if __name__ == '__main__':
  import json
  import sys
  s = read_summary(sys.argv[1])
  s['buffers'] = {'count': len(s['buffers']), 'bytes': sum(b[0] for b in s['buffers']),
                  'initialised_bytes': sum(b[0] for b in s['buffers'] if b[1])}
  print(json.dumps(s, indent=1))
