import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # The CPU oracle is what the suite spends its time in, and torch's default of one thread per core is the wrong setting
  # for it on the GPU boxes (256 logical cores, default 128 threads): measured with scripts/oracle_threads.py (r06), the
  # fp32 train step of efficientdet-d0 at 128 px takes 12.3 s with 128 threads, 1.8 s with 64, 0.59 s with 32, 0.24 s
  # with 16 and 0.18 s with 8; d7x at 768 px forward 7.8 / 2.7 / 1.4 / 0.73 / 0.90 s.  EDET_TEST_THREADS overrides.
  import torch
  want = int(os.environ.get('EDET_TEST_THREADS', '16'))
  torch.set_num_threads(max(1, min(want, torch.get_num_threads())))


def pytest_collection_modifyitems(config, items):
  """Tests marked gpu are skipped (not failed) where there is no HIP device or the library has not been built, so
  a plain `pytest tests` on a CPU machine shows the host-side tests."""
  gpu_items = [it for it in items if it.get_closest_marker('gpu')]
  if not gpu_items:
    return
  import torch
  from automl_amd import _lib
  reason = None
  if not torch.cuda.is_available():
    reason = 'needs an MI355X (no HIP device visible)'
  elif not os.path.exists(_lib.LIB_PATH):
    reason = 'libedet_hip.so has not been built'
  if reason:
    skip = pytest.mark.skip(reason=reason)
    for it in gpu_items:
      it.add_marker(skip)
