"""CPU oracle of the detection post-processing path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/ (and the golden generators under tests/golden/) may import this module.  Plain numpy restatement of
(paths relative to the reference root):

  efficientdet/tf2/postprocess.py   merge_class_box_level_outputs :67-79, topk_class_boxes :82-117, pre_nms :120-157,
                                    nms :160-206, postprocess_global :375-406, per_class_nms :409-467,
                                    postprocess_per_class :470-490, generate_detections_from_nms_output :493-527,
                                    generate_detections :530-586, clip_boxes :61-64
  efficientdet/tf2/anchors.py       decode_box_outputs :30-58
  efficientdet/nms_np.py            hard_nms :84-120, soft_nms :123-184, nms :187-211, per_class_nms :214-265

PARITY STATUS.  nms_np.py is pure numpy and RUNS in the build container: tests/golden/make_golden_postprocess.py
executes it (and the glue of postprocess.py on the torch-backed stand-in) and this module's np_* functions are pinned
against those outputs.  tf.raw_ops.NonMaxSuppressionV5 lives in the un-vendored dependency tensorflow>=2.10,<2.16
(efficientdet/requirements.txt:8; kernel tensorflow/core/kernels/image/non_max_suppression_op.cc,
DoNonMaxSuppressionOp): `nms_v5` restates that published algorithm (score-ordered priority queue with lazy
re-evaluation, suppress_begin_index, IoU with degenerate-area guard, soft-NMS weight exp(-0.5 / sigma * iou^2)) --
parity of nms_v5 versus the TensorFlow binary is UNPINNED.
"""
import heapq

import numpy as np

CLASS_OFFSET = 1               # postprocess.py:26
DUMMY_DETECTION_SCORE = -1e5   # nms_np.py:22


def parse_image_size(image_size):
  """utils.parse_image_size (utils.py:484-506): int, 'WxH' string or (height, width) pair -> (height, width)."""
  if isinstance(image_size, int):
    return (image_size, image_size)
  if isinstance(image_size, str):
    w, h = image_size.lower().split('x')       # the string form is WIDTH x HEIGHT (utils.py:497-499)
    return (int(h), int(w))
  return tuple(int(v) for v in image_size)


def decode_box_outputs(pred_boxes, anchor_boxes):
  """anchors.decode_box_outputs (tf2/anchors.py:30-58), float32 arithmetic as the TF graph does."""
  pred_boxes = np.asarray(pred_boxes, np.float32)
  a = np.asarray(anchor_boxes, np.float32)
  two = np.float32(2)
  ycenter_a = (a[..., 0] + a[..., 2]) / two
  xcenter_a = (a[..., 1] + a[..., 3]) / two
  ha = a[..., 2] - a[..., 0]
  wa = a[..., 3] - a[..., 1]
  ty, tx, th, tw = (pred_boxes[..., i] for i in range(4))
  w = np.exp(tw) * wa
  h = np.exp(th) * ha
  ycenter = ty * ha + ycenter_a
  xcenter = tx * wa + xcenter_a
  return np.stack([ycenter - h / two, xcenter - w / two, ycenter + h / two, xcenter + w / two], axis=-1)


def merge_class_box_level_outputs(num_classes, cls_outputs, box_outputs):
  """postprocess.py:67-79: per-level [B,H,W,A*C] / [B,H,W,A*4] -> [B,N,C] / [B,N,4] (level-major, then y, x, a)."""
  b = cls_outputs[0].shape[0]
  return (np.concatenate([np.asarray(c).reshape(b, -1, num_classes) for c in cls_outputs], 1),
          np.concatenate([np.asarray(x).reshape(b, -1, 4) for x in box_outputs], 1))


def topk_class_boxes(params, cls_outputs, box_outputs):
  """postprocess.py:82-117.  max_nms_inputs > 0: top-k over the flattened (anchor, class) logits (returned in
  descending score order, ties by lower flat index, as tf.math.top_k does); otherwise the per-anchor max class."""
  b, n, c = cls_outputs.shape
  k = params['nms_configs'].get('max_nms_inputs', 0)
  if k > 0:
    flat = cls_outputs.reshape(b, -1)
    top = np.stack([np.argsort(-flat[i], kind='stable')[:k] for i in range(b)])
    indices, classes = top // c, top % c
    cls_topk = np.take_along_axis(flat, top, 1)
    box_topk = np.stack([box_outputs[i][indices[i]] for i in range(b)])
  else:
    classes = np.argmax(cls_outputs, -1).astype(np.int32)
    indices = np.tile(np.arange(n)[None], (b, 1))
    cls_topk = cls_outputs.max(-1)
    box_topk = box_outputs
  return cls_topk, box_topk, classes.astype(np.int32), indices.astype(np.int32)


def sigmoid(x):
  x = np.asarray(x, np.float32)
  return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def pre_nms(params, cls_outputs, box_outputs, anchor_boxes, topk=True):
  """postprocess.py:120-157 -> (boxes [B,K,4], scores [B,K], classes [B,K] or None)."""
  cls_all, box_all = merge_class_box_level_outputs(params['num_classes'], cls_outputs, box_outputs)
  cls_all, box_all = cls_all.astype(np.float32), box_all.astype(np.float32)
  if topk:
    cls_all, box_all, classes, indices = topk_class_boxes(params, cls_all, box_all)
    anchors_sel = np.asarray(anchor_boxes, np.float32)[indices]
  else:
    anchors_sel, classes = np.asarray(anchor_boxes, np.float32), None
  return decode_box_outputs(box_all, anchors_sel), sigmoid(cls_all), classes


# ---------------------------------------------------------------------------------------- TF NonMaxSuppressionV5
def _iou_tf(boxes, i, j):
  """IOU of non_max_suppression_op.cc: corners in any order, 0 for a degenerate box, float32."""
  f = np.float32
  bi, bj = boxes[i], boxes[j]
  ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
  ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
  ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
  ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
  area_i = f(ymax_i - ymin_i) * f(xmax_i - xmin_i)
  area_j = f(ymax_j - ymin_j) * f(xmax_j - xmin_j)
  if area_i <= 0 or area_j <= 0:
    return f(0)
  ih = max(f(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f(0))
  iw = max(f(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f(0))
  inter = f(ih * iw)
  return f(inter / f(f(area_i + area_j) - inter))


def nms_v5(boxes, scores, max_output_size, iou_threshold, score_threshold, soft_nms_sigma, pad_to_max_output_size):
  """tf.raw_ops.NonMaxSuppressionV5 -> (selected_indices, selected_scores, valid_outputs).  See the module header:
  a restatement of DoNonMaxSuppressionOp (float32), not the TensorFlow binary."""
  f = np.float32
  boxes = np.asarray(boxes, np.float32)
  scores = np.asarray(scores, np.float32)
  is_soft = soft_nms_sigma > 0
  scale = f(-0.5) / f(soft_nms_sigma) if is_soft else f(0)
  thr, sthr = f(iou_threshold), f(score_threshold)
  # std::priority_queue with  a < b  <=>  a.score < b.score || (a.score == b.score && a.box_index > b.box_index)
  heap = [(-float(s), i, 0, f(s)) for i, s in enumerate(scores) if s > sthr]
  heapq.heapify(heap)
  selected, selected_scores = [], []
  while len(selected) < max_output_size and heap:
    _, idx, begin, score = heapq.heappop(heap)
    original = score
    hard = False
    for j in range(len(selected) - 1, begin - 1, -1):
      sim = _iou_tf(boxes, idx, selected[j])
      weight = f(np.exp(f(scale * f(sim * sim))))
      score = f(score * (weight if (is_soft or sim <= thr) else f(0)))
      if not is_soft and sim > thr:
        hard = True
        break
      if score <= sthr:
        break
    begin = len(selected)
    if not hard:
      if score == original:
        selected.append(idx)
        selected_scores.append(score)
        continue
      if score > sthr:
        heapq.heappush(heap, (-float(score), idx, begin, score))
  valid = len(selected)
  if pad_to_max_output_size:
    selected = selected + [0] * (max_output_size - valid)
    selected_scores = selected_scores + [f(0)] * (max_output_size - valid)
  return np.asarray(selected, np.int32), np.asarray(selected_scores, np.float32), valid


def nms_thresholds(nms_configs):
  """postprocess.nms :176-191 -> (sigma, iou_thresh, score_thresh) of the TF path."""
  method = nms_configs['method']
  if method == 'hard' or not method:
    return 0.0, nms_configs['iou_thresh'] or 0.5, nms_configs['score_thresh'] or float('-inf')
  if method == 'gaussian':
    return nms_configs['sigma'] or 0.5, 0.5, nms_configs['score_thresh'] or 0.001
  raise ValueError('Inference has invalid nms method {}'.format(method))


def nms(params, boxes, scores, classes, padded):
  """postprocess.nms :160-206 for one image -> (boxes, scores, classes as float, valid_len)."""
  cfg = params['nms_configs']
  sigma, iou_thresh, score_thresh = nms_thresholds(cfg)
  idx, nms_scores, valid = nms_v5(boxes, scores, cfg['max_output_size'], iou_thresh, score_thresh, sigma / 2, padded)
  boxes = np.asarray(boxes, np.float32)
  return boxes[idx].reshape(-1, 4), nms_scores, (np.asarray(classes)[idx] + CLASS_OFFSET).astype(np.float32), valid


def clip_boxes(boxes, image_size):
  h, w = parse_image_size(image_size)
  return np.clip(boxes, np.float32(0), np.asarray([h, w, h, w], np.float32))


def postprocess_global(params, cls_outputs, box_outputs, anchor_boxes, image_scales=None):
  """postprocess.py:375-406 -> (boxes [B,M,4], scores [B,M], classes [B,M], valid_len [B])."""
  boxes, scores, classes = pre_nms(params, cls_outputs, box_outputs, anchor_boxes)
  outs = [nms(params, boxes[i], scores[i], classes[i], True) for i in range(boxes.shape[0])]
  nb, ns, nc, nv = (np.stack([o[k] for o in outs]) for k in range(4))
  nb = clip_boxes(nb, params['image_size'])
  if image_scales is not None:
    nb = nb * np.asarray(image_scales, np.float32)[:, None, None]
  return nb.astype(np.float32), ns, nc, nv.astype(np.int32)


def per_class_nms(params, boxes, scores, classes, image_scales=None):
  """postprocess.py:409-467: V5 per class (unpadded), concatenated in class order, zero-padded, top-k by score."""
  m = params['nms_configs'].get('max_output_size', 100)
  outs = []
  for i in range(boxes.shape[0]):
    bs, ss, cs, vs = [], [], [], []
    for cid in range(params['num_classes']):
      sel = np.where(classes[i] == cid)[0]
      if sel.shape[0] == 0:
        continue
      b, s, c, v = nms(params, boxes[i][sel], scores[i][sel], classes[i][sel], False)
      bs.append(b); ss.append(s); cs.append(c); vs.append(v)
    bs = np.concatenate(bs + [np.zeros((m, 4), np.float32)], 0)
    ss = np.concatenate(ss + [np.zeros((m,), np.float32)], 0)
    cs = np.concatenate(cs + [np.zeros((m,), np.float32)], 0)
    top = np.argsort(-ss, kind='stable')[:m]          # tf.math.top_k: descending, ties by lower index
    outs.append((bs[top], ss[top], cs[top], min(m, int(np.sum(vs)))))
  nb, ns, nc, nv = (np.stack([o[k] for o in outs]) for k in range(4))
  if image_scales is not None:
    nb = nb * np.asarray(image_scales, np.float32)[:, None, None]
  return nb.astype(np.float32), ns, nc, nv.astype(np.int32)


def postprocess_per_class(params, cls_outputs, box_outputs, anchor_boxes, image_scales=None):
  boxes, scores, classes = pre_nms(params, cls_outputs, box_outputs, anchor_boxes)
  return per_class_nms(params, boxes, scores, classes, image_scales)


# ---------------------------------------------------------------------------------------- nms_np.py (numpy NMS)
def _np_overlaps(x1, y1, x2, y2, areas, i, rest):
  """IoU of box i with boxes `rest` in the '+1 pixel' convention of nms_np.py:104-112 (float arithmetic of dets)."""
  xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
  xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
  w = np.maximum(0.0, xx2 - xx1 + 1)
  h = np.maximum(0.0, yy2 - yy1 + 1)
  inter = w * h
  return inter / (areas[i] + areas[rest] - inter)


def np_hard_nms(dets, iou_thresh=None):
  """nms_np.hard_nms :84-120; dets [n,5] = x1, y1, x2, y2, score."""
  iou_thresh = iou_thresh or 0.5
  x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
  areas = (x2 - x1 + 1) * (y2 - y1 + 1)
  order = scores.argsort()[::-1]
  keep = []
  while order.size > 0:
    i = order[0]
    keep.append(i)
    overlap = _np_overlaps(x1, y1, x2, y2, areas, i, order[1:])
    order = order[np.where(overlap <= iou_thresh)[0] + 1]
  return dets[keep]


def np_soft_nms(dets, nms_configs):
  """nms_np.soft_nms :123-184 (linear / gaussian / hard by weights), eager: the current maximum is swapped to the
  front, every remaining score is multiplied by its weight, rows below score_thresh are dropped."""
  method = nms_configs['method']
  sigma = nms_configs['sigma'] or 0.5
  iou_thresh = nms_configs['iou_thresh'] or 0.3
  score_thresh = nms_configs['score_thresh'] or 0.001
  x1, y1, x2, y2 = (dets[:, i] for i in range(4))
  areas = (x2 - x1 + 1) * (y2 - y1 + 1)
  dets = np.concatenate((dets, areas[:, None]), axis=1)
  retained = []
  while dets.size > 0:
    mx = np.argmax(dets[:, 4], axis=0)
    dets[[0, mx], :] = dets[[mx, 0], :]
    retained.append(dets[0, :-1].copy())
    xx1 = np.maximum(dets[0, 0], dets[1:, 0]); yy1 = np.maximum(dets[0, 1], dets[1:, 1])
    xx2 = np.minimum(dets[0, 2], dets[1:, 2]); yy2 = np.minimum(dets[0, 3], dets[1:, 3])
    w = np.maximum(xx2 - xx1 + 1, 0.0)
    h = np.maximum(yy2 - yy1 + 1, 0.0)
    inter = w * h
    iou = inter / (dets[0, 5] + dets[1:, 5] - inter)
    if method == 'linear':
      weight = np.ones_like(iou)
      weight[iou > iou_thresh] -= iou[iou > iou_thresh]
    elif method == 'gaussian':
      weight = np.exp(-(iou * iou) / sigma)
    else:
      weight = np.ones_like(iou)
      weight[iou > iou_thresh] = 0
    dets[1:, 4] *= weight
    dets = dets[np.where(dets[1:, 4] >= score_thresh)[0] + 1, :]
  return np.vstack(retained)


def np_nms(dets, nms_configs):
  """nms_np.nms :187-211 ('diou' is not restated: no configuration of the reference selects it)."""
  method = nms_configs['method']
  if method == 'hard' or not method:
    return np_hard_nms(dets, nms_configs['iou_thresh'])
  if method in ('linear', 'gaussian'):
    return np_soft_nms(dets, nms_configs)
  raise ValueError('Unknown NMS method: {}'.format(method))


def np_per_class_nms(boxes, scores, classes, image_id, image_scale, num_classes, max_boxes_to_draw, nms_configs):
  """nms_np.per_class_nms :214-265 -> [max_boxes_to_draw, 7] = image_id, x1, y1, x2, y2 (scaled), score, class."""
  boxes = boxes[:, [1, 0, 3, 2]]
  detections = []
  for c in range(num_classes):
    sel = np.where(classes == c)[0]
    if sel.shape[0] == 0:
      continue
    top = np_nms(np.column_stack((boxes[sel, :], scores[sel])), nms_configs)
    detections.append(np.column_stack((np.repeat(image_id, len(top)), top, np.repeat(c + 1, len(top)))))

  def dummy(n):
    d = np.zeros((n, 7), dtype=np.float32)
    d[:, 0] = image_id[0]
    d[:, 5] = DUMMY_DETECTION_SCORE
    return d
  if detections:
    detections = np.vstack(detections)
    order = np.argsort(-detections[:, -2])
    detections = np.array(detections[order[0:max_boxes_to_draw]], dtype=np.float32)
    detections = np.vstack([detections, dummy(max(max_boxes_to_draw - len(detections), 0))])
  else:
    detections = dummy(max_boxes_to_draw)
  detections[:, 1:5] *= image_scale
  return detections


def generate_detections(params, cls_outputs, box_outputs, anchor_boxes, image_scales, image_ids, flip=False,
                        per_class=True):
  """postprocess.generate_detections :530-586 -> [B, M, 7] = id, x1, y1, x2, y2, score, class."""
  _, width = parse_image_size(params['image_size'])
  image_scales = np.asarray(image_scales, np.float32)
  image_ids = np.asarray(image_ids)
  widths = image_scales[:, None] * np.float32(width)
  if params['nms_configs'].get('pyfunc', True):
    boxes, scores, classes = pre_nms(params, cls_outputs, box_outputs, anchor_boxes)
    out = []
    for i in range(boxes.shape[0]):
      d = np_per_class_nms(boxes[i], scores[i], classes[i], image_ids[i:i + 1], image_scales[i:i + 1],
                           params['num_classes'], params['nms_configs']['max_output_size'], params['nms_configs'])
      if flip:
        d = np.stack([d[:, 0], widths[i] - d[:, 3], d[:, 2], widths[i] - d[:, 1], d[:, 4], d[:, 5], d[:, 6]], -1)
      out.append(d.astype(np.float32))
    return np.stack(out, 0)
  fn = postprocess_per_class if per_class else postprocess_global
  nb, ns, nc, _ = fn(params, cls_outputs, box_outputs, anchor_boxes, image_scales)
  ids = image_ids.astype(np.float32)[:, None] * np.ones_like(ns)
  if flip:
    cols = [ids, widths - nb[:, :, 3], nb[:, :, 0], widths - nb[:, :, 1], nb[:, :, 2], ns, nc]
  else:
    cols = [ids, nb[:, :, 1], nb[:, :, 0], nb[:, :, 3], nb[:, :, 2], ns, nc]
  return np.stack(cols, -1).astype(np.float32)
