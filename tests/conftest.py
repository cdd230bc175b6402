import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
