"""Headline benchmark: EfficientDet-D0 640x640 forward+backward (one full train_step) images/sec.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run with one rank per GPU; started WITHOUT it, `--gpus N` spawns the N ranks
  itself through `python -m torch.distributed.run --standalone`-style arguments on 127.0.0.1)

One step = forward (training BatchNorm), focal+Huber loss, backward, L2, clip, [gradient all-reduce
SUM over RCCL], SGD-momentum + EMA update on a synthetic COCO-shaped batch already resident in HBM
(BASELINE.json configs[2]; weak scaling: 128 images per GPU).  By default the timed steps REPLAY the step
captured as a hipGraph (--graph 0 / EDET_GRAPH=0: eager launches).  Rank 0 prints ONE JSON line with
`roofline` (dominant kernel family, algorithmic bytes per SURVEY.md section 8d, per-launch HIP events on the
launch stream: inside the timed region in eager mode, over the same K steps repeated eagerly right after it in
graph mode -- a replayed graph has no host-side launch to bracket; `traffic` from the committed PMC passes
under profiles/) and, at N = 1, `cpu_baseline` (the fp32 CPU oracle -- a port, not the reference's TensorFlow
binary, which cannot be installed here -- on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from automl_amd import _lib, hparams_config, netspec, train_lib  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
# algorithmic work per image, D0 640x640 fwd+bwd
ALG_GFLOP_PER_IMG = 23.36
ALG_MB_PER_IMG = 925.0

# C-ABI entry point -> substrings of the kernel names it launches (for the PMC traffic lookup)
ENTRY_KERNELS = {
    'edet_dw_bwd_data': ['dwm::k_dgrad', 'k_dw_bwd_data'],
    'edet_dw_bwd_weight': ['dwm::k_wgrad', 'k_dw_bwd_weight'],
    'edet_dw_fwd': ['dwm::k_fwd', 'k_dw_fwd'],
    'edet_dw_bwd': ['dwm::k_bwd_one', 'dwm::k_bwd_fused'],
    'edet_pw_bwd_weight': ['pws::k_pw_wgrad', 'pwb::k_big_wgrad', 'k_wgrad<unsigned short'],
    'edet_pw_bwd_data': ['pws::k_pw_dgrad', 'pwb::k_big_gemm<true', 'k_gemm<unsigned short, 8, true', 'k_gemm<unsigned short, 4, true', 'k_gemm<unsigned short, 2, true'],
    'edet_pw_bwd': ['pwt::k_pw_bwd_tile', 'pwt::k_gate_finish', 'pws::k_pw_bwd_fused', 'pws::k_noy_apply', 'pws::k_pw_wgrad', 'pwb::k_big_wgrad', 'k_wgrad<unsigned short', 'pws::k_pw_dgrad',
                    'pwb::k_big_gemm<true', 'k_gemm<unsigned short, 8, true', 'k_gemm<unsigned short, 4, true',
                    'k_gemm<unsigned short, 2, true'],
    'edet_pw_fwd': ['pws::k_pw_fwd', 'pwb::k_big_gemm<false', 'k_gemm<unsigned short, 8, false', 'k_gemm<unsigned short, 4, false'],
}


def pmc_traffic(entry, launches_per_step):
  """HBM bytes per launch of `entry` from the newest committed PMC passes (profiles/*_traffic.json, made by
  scripts/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE rocprofv3 runs of this same workload).
  Returns (bytes_per_launch, source) or (None, None): PMC counters cannot be read from inside this process."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))
  pats = ENTRY_KERNELS.get(entry)
  if not files or not pats or not launches_per_step:
    return None, None
  d = json.load(open(files[-1]))
  total = sum(v['fetch_bytes'] + v['write_bytes'] for k, v in d['kernels'].items() if any(p in k for p in pats))
  if total <= 0:
    return None, None
  return total / (d['steps_in_run'] * launches_per_step), os.path.basename(files[-1])


def synth_batch(config, batch, size, seed, device, tdtype):
  """Images N(0,1); ~100 positive anchors / image, ~50 ignored."""
  rng = np.random.default_rng(seed)
  spec = netspec.NetSpec(config)
  fs = spec.feat_sizes(size)
  na = spec.num_anchors
  images = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
  images = images.to(device=device, dtype=tdtype).contiguous()
  labels = {}
  total = sum(fs[l]['height'] * fs[l]['width'] * na for l in range(config.min_level, config.max_level + 1))
  for level in range(config.min_level, config.max_level + 1):
    h, w = fs[level]['height'], fs[level]['width']
    cnt = h * w * na
    ct = np.full((batch, cnt), -1, np.int32)
    bt = np.zeros((batch, cnt, 4), np.float32)
    npos = max(1, int(round(100.0 * cnt / total)))
    nign = max(0, int(round(50.0 * cnt / total)))
    for b in range(batch):
      idx = rng.choice(cnt, npos + nign, replace=False)
      ct[b, idx[:npos]] = rng.integers(0, config.num_classes, npos)
      ct[b, idx[npos:]] = -2
      bt[b, idx[:npos]] = rng.standard_normal((npos, 4)).astype(np.float32) * 0.2
    labels['cls_targets_%d' % level] = torch.from_numpy(ct.reshape(batch, h, w, na)).to(device)
    labels['box_targets_%d' % level] = torch.from_numpy(bt.reshape(batch, h, w, na * 4)).to(device)
  labels['mean_num_positives'] = torch.full((batch,), 100.0, device=device)
  labels['normalizer'] = 100.0 * batch + 1.0     # host copy of sum(mean_num_positives)+1: no device sync
  return images, labels


def physical_cores():
  """Distinct (socket, core) pairs of /proc/cpuinfo; falls back to the logical count."""
  try:
    pairs, phys, core = set(), None, None
    for line in open('/proc/cpuinfo'):
      if line.startswith('physical id'):
        phys = line.split(':')[1].strip()
      elif line.startswith('core id'):
        core = line.split(':')[1].strip()
      elif not line.strip():
        if phys is not None and core is not None:
          pairs.add((phys, core))
        phys = core = None
    return len(pairs) or os.cpu_count()
  except OSError:
    return os.cpu_count()


def cpu_baseline(config, size, seconds_budget=28.0):
  """The fp32 CPU oracle on the host cores, timed with the reference's recipe (tf2/infer_lib.py:181-207: warm-up
  runs, then the mean of timed runs), on a bounded sample of the workloads SURVEY.md section 8d names:
  (i) efficientdet-d0 512x512 batch 1 inference forward (BASELINE configs[0]), (ii) efficientdet-d0 at the
  benchmark image size, batch 8, one full train step (forward + backward + update).  `value` is (ii)."""
  from oracle import efficientdet_oracle as orc
  spec = netspec.NetSpec(config)
  vals = netspec.init_params(spec, 0)
  # thread count: the best of a few candidates on one 512x512 forward (oneDNN on all 128 threads of the GPU box's
  # host is several times slower on these small convolutions than on 16-32)
  probe = orc.Oracle(config=hparams_config.get_efficientdet_config('efficientdet-d0'),
                     params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})
  xp = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 512, 512, 3)).astype(np.float32))
  all_threads = torch.get_num_threads()
  best = (None, all_threads)
  for cand in sorted({all_threads, min(all_threads, 64), min(all_threads, 32), min(all_threads, 16), min(all_threads, 8)}):
    torch.set_num_threads(cand)
    with torch.no_grad():
      probe.forward(xp, False)
      t0 = time.perf_counter()
      probe.forward(xp, False)
      dt = time.perf_counter() - t0
    if best[0] is None or dt < best[0]:
      best = (dt, cand)
  threads = best[1]
  torch.set_num_threads(threads)

  def fresh(cfg):
    return orc.Oracle(config=cfg, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()})

  # (i) forward, 512x512, batch 1
  c512 = hparams_config.get_efficientdet_config('efficientdet-d0')
  o = fresh(c512)
  x = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 512, 512, 3)).astype(np.float32))
  with torch.no_grad():
    for _ in range(3):
      o.forward(x, False)
    n_fwd, t0 = 0, time.perf_counter()
    while n_fwd < 10 and (n_fwd == 0 or time.perf_counter() - t0 < 0.25 * seconds_budget):
      o.forward(x, False)
      n_fwd += 1
    fwd_dt = (time.perf_counter() - t0) / n_fwd
  # (ii) train step, batch 8: one warm-up step on 2 images, then timed steps inside the budget
  batch = 8
  oracle = fresh(config)
  images, labels = synth_batch(config, batch, size, 3, 'cpu', torch.float32)
  labels = {k: v for k, v in labels.items() if k != 'normalizer'}
  with torch.no_grad():
    oracle.forward(images[:1, :64, :64], False)       # registers the variable list
  state = {}
  orc.train_step(oracle, images[:2], {k: v[:2] for k, v in labels.items()}, state, 0.01, 0.9)     # warm-up
  n_step, t0 = 0, time.perf_counter()
  while n_step < 10 and (n_step == 0 or time.perf_counter() - t0 < 0.6 * seconds_budget):
    orc.train_step(oracle, images, labels, state, 0.01, 0.9)
    n_step += 1
  dt = (time.perf_counter() - t0) / n_step
  torch.set_num_threads(all_threads)
  return {'value': batch / dt, 'unit': 'images/sec', 'cores': threads, 'physical_cores': physical_cores(),
          'kind': 'port',
          'forward_d0_512_b1_images_per_sec': 1.0 / fwd_dt,
          'sample': 'fp32 PyTorch-CPU oracle (a restatement of the reference graph: the reference\'s TensorFlow '
                    'binary is not installable here) on %d threads; value = efficientdet-d0 %dx%d batch %d full '
                    'train step, 1 warm-up (2 images) + %d timed; also d0 512x512 batch 1 inference forward, '
                    '3 warm-up + %d timed' % (threads, size, size, batch, n_step, n_fwd)}


def parity_block(config, size):
  """The bf16 benchmark network on 2 images against the CPU oracles (checker only -- the measured path above never
  touches the oracle): (a) inference-mode forward, per level max |logit error| / max |logit| (class, box) against the
  fp32 oracle and against the oracle that emulates the bf16 storage points; (b) training-mode forward (batch
  statistics), layer by layer with teacher forcing: the worst error of any stored tensor against the emulating
  oracle's value computed from the device's own stored inputs (end to end that mode is ill conditioned in bf16:
  tests/test_oracle_conditioning.py)."""
  from oracle import efficientdet_oracle as orc
  from oracle import problems, teacher_force
  # perturbed variables (the tests' problem): with the plain initialisers every class logit is the -log(99) bias and
  # the class column of this block would be vacuous
  vals = problems.perturbed_params(config, 0)
  rng = np.random.default_rng(5)
  images = torch.from_numpy(rng.standard_normal((2, size, size, 3)).astype(np.float32)).to(torch.bfloat16).float()
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals)
  cls, box = net(images, training=False)
  torch.cuda.synchronize()
  out = {'workload': 'efficientdet-d0 %dx%d batch 2 bf16' % (size, size)}

  def fresh(storage):
    return orc.Oracle(config=config, params={k: torch.from_numpy(v.copy()) for k, v in vals.items()}, storage=storage)
  for storage in ('f32', 'bf16'):
    with torch.no_grad():
      cr, br = fresh(storage).forward(images, False)
    errs = []
    for c, r, b, q in zip(cls, cr, box, br):
      errs.append([float((c.float().cpu() - r).abs().max() / r.abs().max()),
                   float((b.float().cpu() - q).abs().max() / q.abs().max())])
    out['inference_forward_vs_' + ('fp32_oracle' if storage == 'f32' else 'bf16_storage_emulating_oracle')] = errs
  net(images, training=True)
  torch.cuda.synchronize()
  o = fresh('bf16')
  hook = o.hook = teacher_force.TeacherForce(net.engine)
  with torch.no_grad():
    o.forward(images, True)
  out['training_forward_layer_by_layer'] = {'stored_tensors_checked': len(hook.fwd_err),
                                            'worst_rel_err': max(hook.fwd_err.values()),
                                            'worst_tensor': max(hook.fwd_err, key=hook.fwd_err.get)}
  # ... and the logits themselves, per pyramid level (the predict layers' stored outputs of that same teacher-forced pass:
  # the worst tensor above is usually one of these -- a bf16-stored logit carries one rounding of 2^-8 of its range)
  import re
  per_level = {'class': {}, 'box': {}}
  for key, e in hook.fwd_err.items():
    m = re.search(r'(class|box)-predict\D*?l(\d+)', key)
    if m:
      per_level[m.group(1)][int(m.group(2))] = max(e, per_level[m.group(1)].get(int(m.group(2)), 0.0))
  out['training_forward_logits_vs_emulating_oracle_per_level'] = {
      k: [v[l] for l in sorted(v)] for k, v in per_level.items() if v}
  return out


def other_configs():
  """The two other single-GPU BASELINE configurations, measured with the same recipe (not the headline value):
  configs[1] efficientnetv2-s backbone 224x224 batch 256 bf16 forward; the per-GPU leg of configs[4]
  efficientdet-d7x 1536x1536 batch 8 full train step."""
  from automl_amd import effnetv2_model
  out = {}
  net = effnetv2_model.EffNetV2Model('efficientnetv2-s', include_top=False, dtype='bf16')
  images = torch.from_numpy(np.random.default_rng(2).standard_normal((256, 224, 224, 3)).astype(np.float32))
  images = images.to('cuda:0', torch.bfloat16).contiguous()
  eng = net._ensure_engine(256, 224, 224)
  for _ in range(3):
    eng.forward(images, training=False)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    eng.forward(images, training=False)
  g.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 20
  out['efficientnetv2-s 224x224 batch 256 bf16 forward (BASELINE configs[1])'] = {
      'images_per_sec': 256 / dt, 'ms_per_step': dt * 1e3, 'steps': 20,
      'hbm_frac': 49.0e6 * 256 / dt / 1e9 / HBM_PEAK_GBS}        # 49.0 MB / image forward
  del net, eng, g, images
  torch.cuda.empty_cache()
  config = hparams_config.get_efficientdet_config('efficientdet-d7x')
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', seed=0, global_batch_size=8, steps_per_epoch=1000,
                                       use_graph=True)
  eng = net._ensure_engine(8, 1536, 1536)
  images, labels = synth_batch(config, 8, 1536, 3, 'cuda:0', eng.tdtype)
  labels.pop('normalizer')
  for _ in range(3):
    net.train_step((images, labels), sync_loss=False)
  images, labels = net.input_buffers()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5):
    net.train_step((images, labels), sync_loss=False)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 5
  out['efficientdet-d7x 1536x1536 batch 8 bf16 train step, stochastic depth on (BASELINE configs[4], per-GPU leg)'] = {
      'images_per_sec': 8 / dt, 'ms_per_step': dt * 1e3, 'steps': 5,
      'hbm_frac': 38630.0e6 * 8 / dt / 1e9 / HBM_PEAK_GBS}      # 38,630 MB / image fwd+bwd
  net._graph, net.engine = None, None
  net._engines.clear()
  eng._bufs.clear()
  del net, eng, images, labels
  torch.cuda.empty_cache()
  # the inference forward of the headline network (north_star: "inference/training path"; inference BatchNorm, the MBConv
  # heads of the 320x320 / 160x160 maps never store their expanded tensor: csrc/mbconv_fused.hip), hipGraph replay
  from automl_amd import efficientdet_net
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=640')
  inet = efficientdet_net.EfficientDetNet(config=config, dtype='bf16', seed=0)
  eng = inet._ensure_engine(128, 640, 640)
  images = torch.from_numpy(np.random.default_rng(2).standard_normal((128, 640, 640, 3)).astype(np.float32))
  images = images.to('cuda:0', torch.bfloat16).contiguous()
  for _ in range(3):
    eng.forward(images, training=False)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    eng.forward(images, training=False)
  g.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20):
    g.replay()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 20
  out['efficientdet-d0 640x640 batch 128 bf16 inference forward (network only, fp32 logits)'] = {
      'images_per_sec': 128 / dt, 'ms_per_step': dt * 1e3, 'steps': 20,
      'fused_mbconv_heads': bool(eng.fused_mbconv_head)}
  inet.engine = None
  inet._engines.clear()
  eng._bufs.clear()
  del inet, eng, g, images
  torch.cuda.empty_cache()
  # the headline workload in the storage precision that meets the north_star's 1e-3 logit tolerance end to end (fp32
  # activations / gradients, the validation kernels: 16x16x4 fp32 MFMA pointwise, generic depthwise) -- what the
  # tolerance costs next to the bf16 line
  config = hparams_config.get_efficientdet_config('efficientdet-d0')
  config.override('image_size=640')
  net = train_lib.EfficientDetNetTrain(config=config, dtype='f32', seed=0, global_batch_size=128, steps_per_epoch=1000)
  eng = net._ensure_engine(128, 640, 640)
  images, labels = synth_batch(config, 128, 640, 3, 'cuda:0', eng.tdtype)
  for _ in range(2):
    net.train_step((images, labels), sync_loss=False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    net.train_step((images, labels), sync_loss=False)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 3
  out['efficientdet-d0 640x640 batch 128 fp32-storage train step (the precision that meets 1e-3 end to end; eager)'] = {
      'images_per_sec': 128 / dt, 'ms_per_step': dt * 1e3, 'steps': 3,
      'hbm_frac': 2 * ALG_MB_PER_IMG * 1e6 * 128 / dt / 1e9 / HBM_PEAK_GBS}     # fp32: twice the bf16 bytes
  net._graph, net.engine = None, None
  net._engines.clear()
  eng._bufs.clear()
  del net, eng, images, labels
  torch.cuda.empty_cache()
  # the headline step WITHOUT the Python engine in the loop: recorded as a step plan, replayed by the library's own host
  # runtime through the network-level C ABI (include/edet_net.h: edet_train_step; hipGraph replay) -- in a process of its
  # own (scripts/bench_plan.py), which also checks the variables against the Python host's bit for bit
  try:
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scripts', 'bench_plan.py'),
                        '--steps', '20'], capture_output=True, text=True, timeout=420)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    d = json.loads(lines[-1])
    out['efficientdet-d0 640x640 batch 128 bf16 train step through the network-level C ABI (edet_train_step: recorded plan, '
        'C host runtime, hipGraph replay)'] = {
            'images_per_sec': d['images_per_sec'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
            'variables_bit_equal_to_the_python_host': bool(d['first_step_equals_python_host'] and
                                                           [v for k, v in d.items() if k.startswith('params_equal')][0]),
            'plan_file_mb': round(d['plan']['file_bytes'] / 1e6, 1), 'plan_calls_per_step': d['plan']['programs']['train_step']}
  except Exception as e:      # noqa: BLE001 -- a side measurement must not take the line down
    out['network-level C ABI replay'] = {'error': str(e)[:200]}
  return out


def spawn_ranks(args):
  """`python bench.py --gpus N` started by hand: run the N ranks through torch.distributed.run on 127.0.0.1 and
  pass rank 0's JSON line through."""
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  raise SystemExit(subprocess.call(cmd, env=dict(os.environ, EDET_BENCH_SPAWNED='1')))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--batch', type=int, default=128, help='images per GPU')
  ap.add_argument('--image_size', type=int, default=640)
  ap.add_argument('--model', default='efficientdet-d0')
  ap.add_argument('--dtype', default='bf16')
  ap.add_argument('--no_cpu_baseline', action='store_true', help='skip the cpu_baseline and parity legs')
  ap.add_argument('--no_other_configs', action='store_true',
                  help='skip the V2-S / D7x side measurements (they run at N = 1 on the headline workload only)')
  ap.add_argument('--dump_launches', default='', help='write the per-(kernel, shape) launch table of one step here')
  ap.add_argument('--force_dist', action='store_true',
                  help='world size 1 through the data-parallel path all the same: RCCL communicator, the gradient '
                       'all-reduce between the two captured graphs (rehearsal of the multi-GPU launch structure on one GPU)')
  ap.add_argument('--clip_gradients_norm', type=float, default=None,
                  help='override config.clip_gradients_norm (hparams_config.py:220); 0 = no clipping')
  ap.add_argument('--overlap_reduce', action='store_true',
                  help='the gradient all-reduce in buckets under the backward pass (needs --clip_gradients_norm 0: the '
                       'reference clips by the global norm of the local gradient before the reduce, train_lib.py:675-683)')
  ap.add_argument('--graph', type=int, default=int(os.environ.get('EDET_GRAPH', '1')),
                  help='1: the timed steps replay the step captured as a hipGraph; 0: eager launches')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    spawn_ranks(args)
  if args.gpus > 1 and world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  # EDET_BENCH_SAME_DEVICE=1 + EDET_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a 1-GPU box (all ranks
  # on cuda:0, gloo between them); never set for a measurement
  if os.environ.get('EDET_BENCH_SAME_DEVICE') == '1':
    local_rank = 0
  backend = os.environ.get('EDET_BENCH_BACKEND', 'nccl')
  device = 'cuda:%d' % local_rank
  torch.cuda.set_device(device)
  dist = None
  use_dist = world > 1 or args.force_dist
  if use_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)

  config = hparams_config.get_efficientdet_config(args.model)
  config.override('image_size=%d' % args.image_size)
  if args.clip_gradients_norm is not None:
    config.clip_gradients_norm = args.clip_gradients_norm
  net = train_lib.EfficientDetNetTrain(config=config, dtype=args.dtype, device=device, seed=0,
                                       global_batch_size=args.batch * world, use_dist=use_dist,
                                       steps_per_epoch=1000, use_graph=bool(args.graph),
                                       overlap_grad_reduce=args.overlap_reduce and use_dist)
  eng = net._ensure_engine(args.batch, args.image_size, args.image_size)
  images, labels = synth_batch(config, args.batch, args.image_size, 3 + rank, device, eng.tdtype)
  norm_host = labels.pop('normalizer')     # graph mode computes the normalizer on the device instead

  def step():
    lb = labels if net.use_graph else dict(labels, normalizer=norm_host)
    net.train_step((images, lb), sync_loss=False)

  # ---- warm-up (graph mode: step 1 eager, step 2 captures, then replays; at least 3 so that the timed
  # region only replays), then one eager, fully profiled step to find the dominant kernel family
  for i in range(max(args.warmup, 3) if args.graph else args.warmup):
    step()
  if args.graph:      # from here on the synthetic batch lives in the captured step's own input buffers
    images, labels = net.input_buffers()
  torch.cuda.synchronize()
  net.use_graph = False
  _lib.profiler = _lib.Profiler(None)
  step()
  torch.cuda.synchronize()
  net.use_graph = bool(args.graph)
  full = _lib.profiler.summary()
  if args.dump_launches and rank == 0:
    rows = sorted(_lib.profiler.by_shape().items(), key=lambda kv: -kv[1][1])
    with open(args.dump_launches, 'w') as f:
      f.write('%-22s %-22s %5s %10s %10s %9s\n' % ('entry point', 'shape', 'calls', 'total_ms', 'MB/call', 'GB/s'))
      for (name, tag), (n, ms, b) in rows:
        f.write('%-22s %-22s %5d %10.3f %10.1f %9.1f\n' % (name, tag, n, ms, b / n / 1e6,
                                                            b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0))
      f.write('TOTAL kernel ms %.3f\n' % sum(v[1] for v in full.values()))
  _lib.profiler = None
  dominant = max(full.items(), key=lambda kv: kv[1][1])[0]
  kernel_ms_total = sum(v[1] for v in full.values())

  # ---- timed region: exactly K steps between barrier + synchronize.  Eager mode: the dominant kernel is
  # HIP-event timed inside this region.  Graph mode: a replayed graph has no per-launch host call to bracket,
  # so the same K steps are repeated eagerly right after the timed region for the roofline leg.
  _lib.profiler = None if args.graph else _lib.Profiler({dominant})
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  if use_dist and args.graph:
    net.collective_timer = []          # HIP events around every gradient all-reduce of the timed region
  t0 = time.perf_counter()
  enqueue_s = []
  for _ in range(args.steps):
    t_step = time.perf_counter()
    step()
    enqueue_s.append(time.perf_counter() - t_step)
  host_enqueue = time.perf_counter() - t0      # host time to enqueue K steps (launches are asynchronous)
  torch.cuda.synchronize()
  local_elapsed = time.perf_counter() - t0     # this rank's own K steps (before the closing barrier)
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  allreduce_ms = None
  if net.collective_timer:
    allreduce_ms = [a.elapsed_time(b) for a, b in net.collective_timer]
  net.collective_timer = None
  eager_ms_per_step = None
  if args.graph:
    net.use_graph = False
    _lib.profiler = _lib.Profiler({dominant})
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      step()
    torch.cuda.synchronize()
    eager_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
  prof = _lib.profiler.summary()
  _lib.profiler = None
  ranks_seen = 1
  rank_ms = [local_elapsed / args.steps * 1e3] * 2      # slowest / fastest rank's own ms per step
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    t = torch.tensor([local_elapsed, -local_elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rank_ms = [float(t[0].item()) / args.steps * 1e3, -float(t[1].item()) / args.steps * 1e3]
    ones = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # through the same RCCL communicator as the gradients
    ranks_seen = int(ones.item())
  losses = eng.loss_values()
  import zlib
  param_crc = zlib.crc32(eng.params_flat.detach().cpu().numpy().tobytes()) & 0xffffffff      # the variables after every step of this run

  if rank == 0:
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    n_l, ms_l, bytes_l = prof[dominant]
    achieved = bytes_l / (ms_l * 1e-3) / 1e9
    is_headline = args.model == 'efficientdet-d0' and args.image_size == 640 and args.batch == 128 and \
        args.dtype == 'bf16' and args.clip_gradients_norm is None
    traffic, traffic_src = pmc_traffic(dominant, n_l / args.steps) if is_headline else (None, None)
    out = {
        'metric': 'images/sec %s %dx%d fwd+bwd (whole job; per-GPU = value / n_gpus)' % (
            args.model.replace('efficientdet-d', 'EfficientDet-D'), args.image_size, args.image_size),
        'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s %dx%d batch %d/GPU forward+backward+SGD/EMA update (%s; '
                               'DP replicas of it for n_gpus>1)' % (
                                   args.model, args.image_size, args.image_size, args.batch,
                                   'BASELINE configs[2]' if is_headline else
                                   ('BASELINE configs[4] per-GPU leg' if 'd7x' in args.model else 'not a BASELINE config')),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'ranks_seen': ranks_seen,
                   'loss': losses.get('loss'), 'param_crc32': param_crc,
                   'collectives': (dist.get_backend() if dist is not None else None),
                   'launch': ('hipGraph replay of the captured step' + (
                       ' (one graph, all-reduce captured inside)' if (net._graph or {}).get('one_graph') else '')) if args.graph else 'eager',
                   'clip_gradients_norm': config.clip_gradients_norm,
                   'grad_reduce': (None if dist is None else
                                   ('%d buckets on a communication stream under the backward pass' % eng._bucket_no)
                                   if net.overlap_grad_reduce else 'one flat all-reduce after the local clip'),
                   # host time per enqueued step over the K back-to-back steps: from the second step on the graph launch
                   # waits for room in the device queue, so this tracks the DEVICE time; the first step of the region
                   # (empty queue after the synchronize) and the fastest one are what the host itself needs
                   'host_enqueue_ms_per_step': host_enqueue / args.steps * 1e3,
                   'host_enqueue_ms_first_step': enqueue_s[0] * 1e3, 'host_enqueue_ms_min': min(enqueue_s) * 1e3,
                   # first contact with a multi-GPU node: the spread of the ranks' own times and the collective itself
                   'per_rank_ms_per_step': {'max': rank_ms[0], 'min': rank_ms[1]},
                   'allreduce_ms_per_step': ({'mean': sum(allreduce_ms) / len(allreduce_ms), 'max': max(allreduce_ms),
                                              'bytes': int(eng.grads_flat.numel()) * 4}
                                             if allreduce_ms else None),
                   'eager_ms_per_step': eager_ms_per_step},
        'roofline': {
            'bound': 'hbm', 'kernel': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
            'traffic_source': traffic_src,
            'launches_per_step': n_l / args.steps, 'avg_launch_ms': ms_l / n_l,
            'timed_over': ('%d eager steps run right after the timed (graph replay) region' % args.steps)
            if args.graph else 'the timed region',
            'algorithmic_bytes_per_launch': bytes_l / n_l,
            'kernel_time_share': full[dominant][1] / kernel_ms_total,
            'whole_step_hbm_frac': (ALG_MB_PER_IMG * 1e6 * args.batch / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS
            if (args.model == 'efficientdet-d0' and args.image_size == 640) else None,
            'whole_step_mfma_frac': (ALG_GFLOP_PER_IMG * 1e9 * args.batch / (ms_per_step * 1e-3) / 2.5e15)
            if (args.model == 'efficientdet-d0' and args.image_size == 640) else None,
            'per_kernel_ms': {k: round(v[1], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])[:12]},
        },
    }
    if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
      out['cpu_baseline'] = cpu_baseline(config, args.image_size)
      if is_headline:
        out['parity'] = parity_block(config, args.image_size)
    if is_headline and world == 1 and not args.no_other_configs:
      net._graph, net.engine = None, None      # release the headline workload's 36 GB of activation buffers
      net._engines.clear()
      eng._bufs.clear()
      torch.cuda.empty_cache()
      out['other_configs'] = other_configs()
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
