"""Generates tests/golden/reference_postprocess.npz by EXECUTING the reference's post-processing code:

  * efficientdet/nms_np.py (pure numpy: hard_nms, soft_nms, nms, per_class_nms) -- the real thing, unmodified;
  * efficientdet/tf2/postprocess.py (merge_class_box_level_outputs, topk_class_boxes, pre_nms, nms, clip_boxes,
    postprocess_global, per_class_nms, postprocess_per_class, generate_detections[_from_nms_output]) and
    tf2/anchors.py (Anchors, decode_box_outputs) -- unmodified, on a torch-backed `tf` (tests/golden/mini_keras.py
    plus the tensor ops below).  tf.raw_ops.NonMaxSuppressionV5 is TensorFlow's C++ kernel and is supplied by
    oracle/postprocess_oracle.nms_v5 (a restatement; see that module's header), so for the two TF-NMS entry points the
    fixture pins the reference's glue around the op, not the op.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_postprocess.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import mini_keras   # noqa
from mini_keras import T, ns   # noqa
from make_golden_anchors import REF   # noqa
from oracle import postprocess_oracle as porc   # noqa


def add_tensor_ops(tf):
  def top_k(x, k=1, sorted=True, **kw):       # descending, ties by lower index (TF's CPU kernel)
    x = T(x)
    idx = torch.argsort(-x, dim=-1, stable=True)[..., :k]
    return torch.gather(x, -1, idx), idx

  def gather_nd(params, indices, batch_dims=0):
    params, indices = T(params), T(indices).long()
    if batch_dims == 1:
      b = torch.arange(params.shape[0]).view(-1, *([1] * (indices.dim() - 2)))
      if indices.shape[-1] == 1:
        return params[b, indices[..., 0]]
      assert indices.shape[-1] == 2
      return params[b, indices[..., 0], indices[..., 1]]
    assert batch_dims == 0 and indices.shape[-1] == 1
    return params[indices[..., 0]]

  def nms_v5(boxes, scores, max_output_size, iou_threshold, score_threshold, soft_nms_sigma, pad_to_max_output_size):
    idx, sc, valid = porc.nms_v5(T(boxes).numpy(), T(scores).numpy(), int(max_output_size), float(iou_threshold),
                                 float(score_threshold), float(soft_nms_sigma), bool(pad_to_max_output_size))
    return T(torch.from_numpy(idx).long()), T(torch.from_numpy(sc)), T(torch.tensor(valid, dtype=torch.int32))

  def numpy_function(fn, inputs, dtype):
    args = [i.detach().numpy() if torch.is_tensor(i) else i for i in inputs]
    return T(torch.from_numpy(np.asarray(fn(*args), np.float32)))

  def stack(xs, axis=0, name=None):
    return torch.stack([T(torch.as_tensor(x)) for x in xs], dim=axis)

  def pad(x, paddings, **kw):
    flat = []
    for lo, hi in reversed(paddings):
      flat += [int(lo), int(hi)]
    return torch.nn.functional.pad(T(x), flat)

  def clip_by_value(x, lo, hi):
    return torch.minimum(torch.maximum(T(x), torch.tensor(lo, dtype=x.dtype)), torch.tensor(hi, dtype=x.dtype))

  tf.math = ns('math', top_k=top_k, argmax=lambda x, axis=-1, output_type=None: torch.argmax(T(x), dim=axis).int(),
               sigmoid=torch.sigmoid, exp=torch.exp)
  tf.int32, tf.int64 = torch.int32, torch.int64
  tf.gather_nd = gather_nd
  tf.gather = lambda params, indices, **kw: T(params)[T(indices).long()]
  tf.raw_ops = ns('raw_ops', NonMaxSuppressionV5=nms_v5)
  tf.numpy_function = numpy_function
  tf.stack = stack
  tf.pad = pad
  tf.clip_by_value = clip_by_value
  tf.range = lambda n: T(torch.arange(int(n), dtype=torch.int32))
  tf.tile = lambda x, reps: T(x).repeat(*[int(r) for r in reps])
  tf.reduce_max = lambda x, axis=None: T(x).max(dim=axis).values if axis is not None else T(x).max()
  tf.unstack = lambda x, num=None, axis=0: list(torch.unbind(T(x), dim=axis))
  tf.where = lambda cond: torch.nonzero(cond)
  tf.equal = lambda a, b: T(a) == b
  tf.minimum = lambda a, b: torch.minimum(torch.as_tensor(a), torch.as_tensor(b))
  tf.slice = lambda x, begin, size: T(x)[int(begin[0]):int(begin[0]) + int(size[0])]
  tf.ones_like = torch.ones_like
  tf.transpose = lambda x, perm: T(x).permute(*perm)


def make_inputs(rng, batch, sizes, num_anchors, num_classes, hot):
  """Per-level logits / box codes.  A few logits per image are pushed up so that NMS has clusters to resolve."""
  cls, box = [], []
  for s in sizes:
    c = rng.normal(-3.0, 1.5, (batch, s, s, num_anchors * num_classes)).astype(np.float32)
    m = rng.random(c.shape) < hot
    c[m] += rng.uniform(3.0, 7.0, int(m.sum())).astype(np.float32)
    cls.append(c)
    box.append(rng.normal(0.0, 0.25, (batch, s, s, num_anchors * 4)).astype(np.float32))
  return cls, box


def params_for(image_size, min_level, max_level, num_classes, nms_configs):
  return dict(min_level=min_level, max_level=max_level, aspect_ratios=[1.0, 2.0, 0.5], num_scales=3, anchor_scale=4.0,
              image_size=image_size, num_classes=num_classes, data_format='channels_last', nms_configs=nms_configs)


CASES = {   # name: (image_size, min_level, max_level, num_classes, batch, seed, hot fraction)
    'a': (128, 3, 7, 90, 2, 101, 0.002),
    'b': (96, 3, 5, 7, 3, 102, 0.02),
}
NMS_CONFIGS = {
    'gaussian': dict(method='gaussian', iou_thresh=None, score_thresh=0., sigma=None, pyfunc=False, max_nms_inputs=0,
                     max_output_size=100),
    'hard': dict(method='hard', iou_thresh=None, score_thresh=None, sigma=None, pyfunc=False, max_nms_inputs=0,
                 max_output_size=20),
    'gaussian_topk': dict(method='gaussian', iou_thresh=None, score_thresh=0.01, sigma=0.3, pyfunc=False,
                          max_nms_inputs=300, max_output_size=50),
    'linear': dict(method='linear', iou_thresh=0.4, score_thresh=0.02, sigma=None, pyfunc=True, max_nms_inputs=0,
                   max_output_size=30),
}


def main():
  tf = mini_keras.build_tf()
  add_tensor_ops(tf)
  mini_keras.install(tf)
  for stubbed in ('nms_np', 'tf2.postprocess'):      # the real modules, not the import stubs
    sys.modules.pop(stubbed, None)
  sys.path.insert(0, REF)
  import nms_np as ref_nms_np              # noqa: the reference modules
  from tf2 import postprocess as ref_pp    # noqa
  assert ref_pp.nms_np is ref_nms_np and hasattr(ref_nms_np, 'soft_nms')
  out = {}
  for cname, (size, lo, hi, ncls, batch, seed, hot) in CASES.items():
    rng = np.random.default_rng(seed)
    sizes, s = [], size
    for level in range(1, hi + 1):
      s = (s - 1) // 2 + 1
      if level >= lo:
        sizes.append(s)
    cls, box = make_inputs(rng, batch, sizes, 9, ncls, hot)
    scales = rng.uniform(0.5, 2.0, batch).astype(np.float32)
    ids = np.arange(batch) + 7
    for i, (c, b) in enumerate(zip(cls, box)):
      out['%s/cls_%d' % (cname, i)] = c
      out['%s/box_%d' % (cname, i)] = b
    out['%s/scales' % cname] = scales
    out['%s/ids' % cname] = ids
    tcls = lambda: [torch.from_numpy(c) for c in cls]     # noqa: fresh lists (the reference mutates them)
    tbox = lambda: [torch.from_numpy(b) for b in box]     # noqa
    for nname, cfg in NMS_CONFIGS.items():
      params = params_for(size, lo, hi, ncls, dict(cfg))
      key = '%s/%s/' % (cname, nname)
      b_, s_, c_ = ref_pp.pre_nms(params, tcls(), tbox())
      out[key + 'pre_boxes'], out[key + 'pre_scores'], out[key + 'pre_classes'] = (
          b_.numpy(), s_.numpy(), c_.numpy().astype(np.int32))
      if cfg['method'] in ('gaussian', 'hard'):      # the TF-NMS entry points accept these two methods only
        for fn, tag in ((ref_pp.postprocess_global, 'global'), (ref_pp.postprocess_per_class, 'per_class')):
          r = fn(params, tcls(), tbox(), torch.from_numpy(scales))
          for nm, v in zip(('boxes', 'scores', 'classes', 'valid'), r):
            out[key + tag + '_' + nm] = v.numpy()
        out[key + 'det_tf'] = ref_pp.generate_detections(params, tcls(), tbox(), torch.from_numpy(scales),
                                                         torch.from_numpy(ids), False).numpy()
      # the numpy NMS path (nms_configs.pyfunc = True): pre_nms + the REAL nms_np.per_class_nms
      params['nms_configs']['pyfunc'] = True
      for flip in (False, True):
        out[key + 'det_np_%d' % flip] = ref_pp.generate_detections(
            params, tcls(), tbox(), torch.from_numpy(scales), torch.from_numpy(ids), flip).numpy()
  # nms_np on its own: a clustered set of boxes through every method of nms_np.nms
  rng = np.random.default_rng(7)
  centers = rng.uniform(20, 100, (12, 2))
  dets = []
  for cx, cy in centers:
    for _ in range(10):
      w, h = rng.uniform(10, 40, 2)
      dx, dy = rng.normal(0, 4, 2)
      dets.append([cx + dx - w / 2, cy + dy - h / 2, cx + dx + w / 2, cy + dy + h / 2, rng.uniform(0.01, 1.0)])
  dets = np.asarray(dets, np.float32)
  out['np/dets'] = dets
  for method, cfg in (('hard', dict(method='hard', iou_thresh=0.45, score_thresh=None, sigma=None)),
                      ('gaussian', dict(method='gaussian', iou_thresh=None, score_thresh=None, sigma=None)),
                      ('linear', dict(method='linear', iou_thresh=None, score_thresh=0.05, sigma=None))):
    out['np/' + method] = ref_nms_np.nms(dets.copy(), cfg)
  np.savez_compressed(os.path.join(HERE, 'reference_postprocess.npz'), **out)
  print('reference_postprocess.npz', len(out), 'arrays')
  for k in sorted(out):
    if k.endswith('valid') or k.startswith('np/'):
      print(' ', k, out[k].shape, out[k] if k.endswith('valid') else '')


if __name__ == '__main__':
  main()
