"""Builds the gfx950 shared library (C ABI in include/edet_hip.h) in-tree with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libedet_hip.so')
SOURCES = ['pw_gemm.hip', 'pw_stream.hip', 'pw_big.hip', 'pw_glds.hip', 'pw_tile_bwd.hip', 'conv.hip', 'conv_halo.hip', 'dwconv.hip', 'dw_march.hip', 'mbconv_fused.hip', 'stem.hip', 'bn_se.hip', 'fuse.hip', 'loss_opt.hip', 'postprocess.hip', 'labeling.hip', 'preprocess.hip', 'error.cpp', 'net_runtime.cpp']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']
# files that restate float32 numpy / TensorFlow expressions operation by operation (argmax ties, 1e-6 parities):
# no fused multiply-add contraction (hipcc's default is -ffp-contract=fast, and HIP's __fmul_rn / __fadd_rn are
# plain operators, not contraction barriers)
EXTRA_FLAGS = {'labeling.hip': ['-ffp-contract=off'], 'preprocess.hip': ['-ffp-contract=off']}


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
  """Compiles every HIP source for gfx950 and links libedet_hip.so; returns its path."""
  hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(_HERE, '..', 'include', 'edet_hip.h'),
          os.path.join(_HERE, '..', 'include', 'edet_net.h'), os.path.join(CSRC, 'plan_stubs.inc')]
  objdir = os.path.join(CSRC, 'build')
  os.makedirs(objdir, exist_ok=True)
  jobs = []
  for src in SOURCES:
    s = os.path.join(CSRC, src)
    o = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
    if force or _stale(o, [s] + hdrs):
      cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', s, '-o', o]
      jobs.append(cmd)

  def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stderr))
    if verbose and r.stderr:
      sys.stderr.write(r.stderr)

  if jobs:
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
      list(ex.map(run, jobs))
  objs = [os.path.join(objdir, os.path.splitext(s)[0] + '.o') for s in SOURCES]
  if force or jobs or _stale(LIB_PATH, objs):
    run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs)
  return LIB_PATH


if __name__ == '__main__':
  print(build_library(force='--force' in sys.argv, verbose=True))
