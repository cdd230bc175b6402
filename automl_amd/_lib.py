"""ctypes binding of libedet_hip.so (C ABI declared in include/edet_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails this module raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libedet_hip.so')

EDET_F32, EDET_BF16 = 0, 1
ACT_NONE, ACT_SWISH, ACT_RELU, ACT_RELU6, ACT_HSWISH, ACT_MISH, ACT_SRELU = 0, 1, 2, 3, 4, 5, 6
ACT_CODES = {'swish': ACT_SWISH, 'silu': ACT_SWISH, 'swish_native': ACT_SWISH, 'relu': ACT_RELU, 'relu6': ACT_RELU6,
             'hswish': ACT_HSWISH, 'mish': ACT_MISH, 'srelu': ACT_SRELU}
RS_IDENTITY, RS_UP2, RS_POOL = 0, 1, 2
MAX_PARTS = 1024
OPT_SPLIT = 16   # EDET_OPT_SPLIT: partial squared norms per tensor segment
SEG_L2, SEG_FROZEN = 1, 2   # EDET_SEG_L2 / EDET_SEG_FROZEN bits of seg_flags

c_void_p, c_int, c_float, c_double, c_int64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_double, ctypes.c_int64)


class TView(ctypes.Structure):
  _fields_ = [('data', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('gate', c_void_p),
              ('act', c_int), ('n', c_int), ('h', c_int), ('w', c_int), ('c', c_int), ('ld', c_int)]


class GView(ctypes.Structure):
  _fields_ = [('dz', c_void_p), ('y', c_void_p), ('a', c_void_p), ('b', c_void_p), ('cc', c_void_p),
              ('n', c_int), ('h', c_int), ('w', c_int), ('c', c_int), ('ld', c_int)]


class BwdEpi(ctypes.Structure):
  _fields_ = [('gout', c_void_p), ('beta', c_int), ('mean', c_void_p), ('rstd', c_void_p),
              ('stat_partials', c_void_p), ('dgate', c_void_p), ('flags', c_int)]


EPI_Y_IS_CONV_OF_INPUT = 1      # EDET_EPI_Y_IS_CONV_OF_INPUT


class NmsCfg(ctypes.Structure):
  _fields_ = [('method', c_int), ('convention', c_int), ('iou_thresh', c_float), ('score_thresh', c_float),
              ('sigma', c_float), ('max_output_size', c_int)]


NMS_HARD, NMS_GAUSSIAN, NMS_LINEAR = 0, 1, 2
NMS_TF_V5, NMS_NUMPY = 0, 1
NMS_PAD_INDEX0, NMS_PAD_ZERO, NMS_PAD_DUMMY = 0, 1, 2

PT, PG, PE, PI = (ctypes.POINTER(TView), ctypes.POINTER(GView), ctypes.POINTER(BwdEpi),
                  ctypes.POINTER(c_int))

# name -> argtypes; every function returns int.  Must list EVERY symbol of include/edet_hip.h
# (tests/test_abi.py checks the header against this table and the built library).
SIGNATURES = {
    'edet_debug_launch_log': [c_int],
    'edet_debug_launch_names': [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)],
    'edet_cast': [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    'edet_cast_matrix': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'edet_cast_batch': [c_void_p, c_int, c_int, c_int, c_void_p],
    'edet_stem_fwd': [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, PI,
                      c_int, c_void_p],
    'edet_stem_bwd_weight': [c_void_p, c_int, c_int, c_int, PG, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_pw_fwd': [PT, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, PI, c_int, c_void_p],
    'edet_pw_fwd_f32out': [PT, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'edet_pw_bwd_data': [PG, c_void_p, c_int, PT, PE, PI, c_int, c_void_p],
    'edet_pw_bwd_weight': [PT, PG, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_pw_bwd': [PG, c_void_p, c_int, PT, PE, PI, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_conv_fwd': [PT, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, PI, c_int, c_void_p],
    'edet_conv_bwd_data': [PG, c_void_p, c_int, c_int, c_int, PT, PE, PI, c_int, c_void_p],
    'edet_conv_bwd_weight': [PT, PG, c_int, c_int, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_mbconv_fused_supported': [PT, c_int, c_int, c_int, c_int],
    'edet_mbconv_expand_stats': [PT, c_void_p, c_int, c_int, c_void_p, PI, c_int, c_void_p],
    'edet_mbconv_expand_dw_fwd': [PT, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_int, c_void_p, c_int, c_void_p, PI, c_int, c_void_p],
    'edet_dw_fwd': [PT, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, PI, c_int, c_void_p],
    'edet_dw_bwd_data': [PG, c_void_p, c_int, c_int, PT, PE, PI, c_int, c_void_p],
    'edet_dw_bwd_weight': [PT, PG, c_int, c_int, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_dw_bwd': [PG, c_void_p, c_int, c_int, PT, PE, PI, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_reduce_defer': [c_void_p, c_int],
    'edet_reduce_flush': [c_void_p],
    'edet_reduce_deferred_end': [c_void_p, ctypes.POINTER(c_void_p)],
    'edet_bn_finalize': [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_float, c_float, c_int,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_bn_eval': [c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_bn_bwd_reduce': [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, PI,
                           c_int, c_void_p],
    'edet_bn_bwd_finalize': [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_bn_res': [PT, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'edet_add': [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p],
    'edet_se_pool': [PT, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_se_squeeze_excite': [PT, c_void_p, ctypes.c_size_t, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'edet_se_fc': [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                   c_void_p, c_void_p, c_int, c_void_p],
    'edet_se_fc_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    'edet_se_gate_bwd': [PT, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, PI, c_int, c_void_p],
    'edet_fuse_weights': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    'edet_fuse_fwd': [PT, PT, PT, PI, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                      ctypes.POINTER(c_void_p), c_int, c_int, c_void_p],
    'edet_fuse_bwd_pre': [PT, PT, PT, PI, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                          c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), PI, c_int, c_void_p, ctypes.c_size_t,
                          ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_void_p), c_int, c_void_p],
    'edet_fuse_bwd_input': [PT, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                            c_int, c_int, c_void_p],
    'edet_fuse_weights_bwd': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p],
    'edet_focal_loss': [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_focal_loss_smooth': [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_box_loss': [c_void_p, c_int, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_void_p,
                      c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p],
    'edet_opt_l2_norms': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p],
    'edet_opt_clip_factors': [c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_opt_scale': [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    'edet_opt_sgd_ema': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                         c_float, c_void_p],
    'edet_opt_adam_ema': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                          c_float, c_float, c_float, c_void_p],
    'edet_zero': [c_void_p, ctypes.c_size_t, c_void_p],
    'edet_compact_rows': [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p],
    'edet_cast_to_f32': [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    'edet_axpy_clear': [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    'edet_loss_normalizer': [c_void_p, c_int, c_void_p, c_void_p],
    'edet_pre_nms': [c_void_p, c_void_p, PI, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                     c_void_p, c_void_p],
    'edet_pre_nms_topk_workspace_bytes': [c_int, c_int, ctypes.POINTER(ctypes.c_size_t)],
    'edet_pre_nms_topk': [c_void_p, c_void_p, PI, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                          ctypes.c_size_t, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_nms_workspace_bytes': [c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_size_t)],
    'edet_nms': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(NmsCfg), c_void_p,
                 ctypes.c_size_t, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_label_anchors_workspace_bytes': [c_int, c_int, ctypes.POINTER(ctypes.c_size_t)],
    'edet_label_anchors': [c_void_p, PI, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                           ctypes.c_size_t, c_void_p, c_void_p, c_void_p, c_void_p],
    'edet_preprocess_infer': [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p],
    'edet_preprocess_train': [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    'edet_nms_gather': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
}

_lib = None


class EdetError(RuntimeError):
  pass


def load():
  """Loads the shared library (once). Raises if it has not been built."""
  global _lib
  if _lib is not None:
    return _lib
  # torch bundles its own libamdhip64.so (soname libamdhip64.so.7, the soname this library needs): it
  # must be in the process BEFORE libedet_hip.so is dlopen'ed, otherwise the system HIP runtime gets
  # loaded next to torch's, and the second runtime to initialise sees no device.
  import torch  # noqa: F401
  if not os.path.exists(LIB_PATH):
    raise EdetError(
        'libedet_hip.so not found at %s: the gfx950 HIP library has not been built '
        '(run `python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.' % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  lib.edet_last_error.restype = ctypes.c_char_p
  lib.edet_last_error.argtypes = []
  lib.edet_version.restype = c_int
  lib.edet_version.argtypes = []
  for name, argtypes in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = c_int
    fn.argtypes = argtypes
  _lib = lib
  return lib


class Profiler(object):
  """Per-launch HIP-event timing on the launch stream (bench.py's roofline leg).

  names: None = every entry point, else a set of entry-point names to time.  Each record is
  (name, algorithmic_bytes, start_event, end_event, shape_tag); events are recorded on torch's current
  stream, which is the stream every kernel of this library is launched on.
  """

  def __init__(self, names=None):
    self.names = names
    self.records = []

  def summary(self):
    """name -> (launches, total_ms, total_bytes); call after torch.cuda.synchronize()."""
    out = {}
    for name, nbytes, s, e, _ in self.records:
      n, ms, b = out.get(name, (0, 0.0, 0))
      out[name] = (n + 1, ms + s.elapsed_time(e), b + nbytes)
    return out

  def by_shape(self):
    """(name, tag) -> (launches, total_ms, total_bytes), for the per-layer launch table."""
    out = {}
    for name, nbytes, s, e, tag in self.records:
      n, ms, b = out.get((name, tag), (0, 0.0, 0))
      out[(name, tag)] = (n + 1, ms + s.elapsed_time(e), b + nbytes)
    return out


profiler = None
recorder = None      # automl_amd.plan.Recorder while a step plan is being recorded: sees every call before it is made


def call(name, *args, nbytes=0, tag=''):
  """Calls lib.<name>(*args); raises EdetError with edet_last_error() on failure."""
  lib = load()
  if recorder is not None:
    recorder.on_call(name, args)
  p = profiler
  if p is not None and (p.names is None or name in p.names):
    import torch
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = getattr(lib, name)(*args)
    e.record()
    p.records.append((name, nbytes, s, e, tag))
  else:
    rc = getattr(lib, name)(*args)
  if rc != 0:
    raise EdetError('%s failed (%d): %s' % (name, rc, lib.edet_last_error().decode()))


def launch_log_start():
  """Clears and starts the library's debug launch log (kernel symbol -> launches)."""
  call('edet_debug_launch_log', 1)


def launch_log_stop():
  """Stops the log and returns {demangled kernel name: launches}."""
  call('edet_debug_launch_log', 0)
  need = ctypes.c_size_t(0)
  call('edet_debug_launch_names', None, 0, ctypes.byref(need))
  buf = ctypes.create_string_buffer(need.value + 1)
  call('edet_debug_launch_names', buf, need.value + 1, ctypes.byref(need))
  out = {}
  for line in buf.value.decode().splitlines():
    count, name = line.split('\t', 1)
    out[name] = int(count)
  return out


def ptr(t):
  """Device/host pointer of a torch tensor (None -> NULL)."""
  return None if t is None else t.data_ptr()
