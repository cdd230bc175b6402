#!/usr/bin/env python3
"""The training step of bench.py (EfficientDet-D0 640x640, batch 128, bf16) run WITHOUT the Python engine in the loop:
recorded once as a step plan (automl_amd/plan.py), loaded by the library's host runtime (include/edet_net.h) and replayed as
one captured hipGraph per step through edet_train_step.  Prints one JSON line: ms per replayed step of the C runtime, the
plan's size and load time, and whether the variables after the replayed steps equal the Python host's bit for bit."""
import argparse
import json
import os
import sys
import tempfile
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from automl_amd import hparams_config, net_c, plan, train_lib  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--model', default='efficientdet-d0')
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--image_size', type=int, default=640)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--plan', default='')
  ap.add_argument('--detect', action='store_true', help='also measure edet_detect (an inference-only plan)')
  args = ap.parse_args()
  config = hparams_config.get_efficientdet_config(args.model)
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', seed=0)
  images, labels = bench.synth_batch(config, args.batch, args.image_size, 3, 'cuda:0', torch.bfloat16)
  labels.pop('normalizer')
  path = args.plan or os.path.join(tempfile.mkdtemp(prefix='edet_plan_'), 'step.plan')
  lr, decay = 0.008, 0.9998
  t0 = time.perf_counter()
  summary, expected = plan.record_network(net, images, labels, path, learning_rate=lr, ema_decay=decay,
                                          max_init_bytes=1 << 20)
  t_record = time.perf_counter() - t0
  eng = net._ensure_engine(args.batch, args.image_size, args.image_size)
  dimages, dl = net._to_device_images(images, eng), net._labels_to_device(labels, eng)
  t0 = time.perf_counter()
  cnet = net_c.CNet(path)
  t_load = time.perf_counter() - t0
  cnet.use_graph(True)
  st = torch.cuda.Stream()
  cnet.train_step(lr, decay, st.cuda_stream)          # = the recorded step (eager)
  st.synchronize()
  same = bool(np.array_equal(cnet.read('params').view(np.uint32), expected['params'].view(np.uint32).reshape(-1)))
  for _ in range(args.warmup):                        # capture + first replays
    cnet.train_step(lr, decay, st.cuda_stream)
  st.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(st)
  for _ in range(args.steps):
    cnet.train_step(lr, decay, st.cuda_stream)
  e1.record(st)
  st.synchronize()
  ms = e0.elapsed_time(e1) / args.steps
  # the Python host takes the same number of further steps from the same state: variables must agree bit for bit
  for _ in range(args.warmup + args.steps):
    plan.train_pass(eng, dimages, dl, lr, decay)
  torch.cuda.synchronize()
  got = cnet.read('params').view(np.uint32)
  want = eng.params_flat.cpu().numpy().view(np.uint32).reshape(-1)
  print(json.dumps({
      'workload': '%s %dx%d batch %d bf16 train step through edet_train_step (C host runtime, hipGraph replay of a recorded plan)'
                  % (args.model, args.image_size, args.image_size, args.batch),
      'ms_per_step': ms, 'images_per_sec': args.batch / ms * 1e3, 'steps': args.steps,
      'plan': {**summary, 'file_bytes': os.path.getsize(path), 'record_s': round(t_record, 2), 'load_s': round(t_load, 2)},
      'first_step_equals_python_host': same,
      'params_equal_python_host_after_%d_steps' % (1 + args.warmup + args.steps): bool(np.array_equal(got, want)),
      'param_crc32': int(zlib.crc32(got.tobytes()))}))
  cnet.close()
  if not args.plan:
    os.remove(path)
  if args.detect:
    # the serving direction: raw uint8 images -> detections through edet_detect (preprocessing + network with inference
    # BatchNorm + pre_nms + global NMS), an inference-only plan, hipGraph replay
    from automl_amd import efficientdet_net
    del net, eng, dimages, dl, expected
    torch.cuda.empty_cache()
    config = hparams_config.get_efficientdet_config(args.model)
    config.override('image_size=%d' % args.image_size)
    inet = efficientdet_net.EfficientDetNet(config=config, dtype='bf16', seed=0)
    rh, rw = args.image_size * 3 // 4, args.image_size
    raw = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (args.batch, rh, rw, 3), dtype=np.uint8))
    dpath = path + '.detect'
    dsummary, dexp = plan.record_network(inet, images, None, dpath, detect_raw_hw=(rh, rw), raw_images=raw,
                                         max_init_bytes=1 << 20)
    dnet = net_c.CNet(dpath)
    dnet.use_graph(True)
    for _ in range(3):
      dnet.detect(st.cuda_stream)
    st.synchronize()
    ok = bool(np.array_equal(dnet.read('detections.boxes').view(np.uint32), dexp['detections.boxes'].view(np.uint32).reshape(-1)))
    e0.record(st)
    for _ in range(args.steps):
      dnet.detect(st.cuda_stream)
    e1.record(st)
    st.synchronize()
    dms = e0.elapsed_time(e1) / args.steps
    print(json.dumps({
        'workload': '%s %dx%d batch %d bf16 detection through edet_detect (raw %dx%d uint8 images -> preprocess -> network -> '
                    'global NMS; C host runtime, hipGraph replay)' % (args.model, args.image_size, args.image_size, args.batch, rh, rw),
        'ms_per_batch': dms, 'images_per_sec': args.batch / dms * 1e3, 'steps': args.steps,
        'plan': {**dsummary, 'file_bytes': os.path.getsize(dpath)}, 'detections_equal_recorded_pass': ok,
        'mean_valid_detections': float(dexp['detections.valid_len'].mean())}))
    dnet.close()
    os.remove(dpath)


if __name__ == '__main__':
  main()
