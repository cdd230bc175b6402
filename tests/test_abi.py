"""The C-ABI shared library loads and exports every symbol include/edet_hip.h declares; the ctypes
table mirrors the header (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from automl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'edet_hip.h')


def header_functions():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  out = {}
  for m in re.finditer(r'\b(?:int|const char\*)\s+(edet_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
    args = [a.strip() for a in m.group(2).split(',') if a.strip() and a.strip() != 'void']
    out[m.group(1)] = args
  return out


def test_header_declares_what_ctypes_binds():
  fns = header_functions()
  assert 'edet_last_error' in fns and 'edet_version' in fns
  declared = set(fns) - {'edet_last_error', 'edet_version'}
  assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
  for name, argtypes in _lib.SIGNATURES.items():
    assert len(argtypes) == len(fns[name]), (name, len(argtypes), fns[name])


def test_library_exports_every_symbol():
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for name in list(header_functions()):
    assert hasattr(lib, name), 'libedet_hip.so does not export %s' % name
  lib.edet_version.restype = ctypes.c_int
  assert lib.edet_version() >= 1


def test_struct_layouts_match_header():
  assert ctypes.sizeof(_lib.TView) == 4 * 8 + 6 * 4
  assert ctypes.sizeof(_lib.GView) == 5 * 8 + 5 * 4 + 4      # padded to 8
  assert ctypes.sizeof(_lib.BwdEpi) == 7 * 8                  # int beta and int flags padded


def test_missing_library_fails_loudly(monkeypatch):
  monkeypatch.setattr(_lib, '_lib', None)
  monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libedet_hip.so')
  with pytest.raises(_lib.EdetError):
    _lib.load()


def test_engine_refuses_to_run_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from automl_amd import efficientdet_net
  net = efficientdet_net.EfficientDetNet('efficientdet-d0')
  with pytest.raises(_lib.EdetError):
    net(torch.zeros(1, 64, 64, 3))


def test_fused_mbconv_kernels_do_not_spill_registers():
  """automl_amd/csrc/mbconv_fused.hip: no instantiation of the fused MBConv head may spill vector registers (r06: the
  six-wave 3x3 stride-2 instantiation, built for four waves per SIMD with 7 spilled VGPRs, stored wrong values in the
  first row of a tile on the device).  hipcc cross-compiles without a GPU; the resource report is the compiler's."""
  import os
  import re
  import subprocess
  import tempfile
  from automl_amd import build
  src = os.path.join(build.CSRC, 'mbconv_fused.hip')
  with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run([build.HIPCC] + build.FLAGS + ['-c', src, '-o', os.path.join(tmp, 'm.o'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  names = re.findall(r'Function Name: (\S+)', r.stderr)
  spills = [int(v) for v in re.findall(r'VGPRs Spill: (\d+)', r.stderr)]
  scratch = [int(v) for v in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', r.stderr)]
  assert len(names) == len(spills) == len(scratch) and len(names) >= 30, (len(names), len(spills))
  bad = [(n, s, c) for n, s, c in zip(names, spills, scratch) if 'k_exp_dw_fwd' in n and (s or c)]
  assert not bad, bad


def test_lds_dma_kernel_keeps_its_prefetch_in_flight():
  """automl_amd/csrc/pw_glds.hip: between an LDS-DMA and the LDS accesses that follow it hipcc inserts `s_waitcnt vmcnt(0)`
  unless the accesses carry an alias scope (the `__restrict__` helpers of that file) -- one merged store or one plain
  dereference in the main loop and the three reduction steps in flight become none, silently (the results stay right).
  Checked in the compiler's own assembly of every instantiation: inside the reduction loop -- from the first counted wait
  to the last hand-written `s_barrier` -- the only `vmcnt` waits are the hand-written ones
  (between ;;#ASMSTART / ;;#ASMEND); and no instantiation spills vector registers."""
  import os
  import re
  import subprocess
  import tempfile
  from automl_amd import build
  src = os.path.join(build.CSRC, 'pw_glds.hip')
  with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run([build.HIPCC] + build.FLAGS + ['-c', src, '-o', os.path.join(tmp, 'g.o'), '-save-temps=obj',
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
    assert len(asm) == 1, os.listdir(tmp)
    text = open(os.path.join(tmp, asm[0])).read()
  spills = [int(v) for v in re.findall(r'VGPRs Spill: (\d+)', r.stderr)]
  assert len(spills) >= 4 and not any(spills), spills
  kernels = re.findall(r'^(_ZN3pwg10k_wide_fwd\w+):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M)
  assert len(kernels) >= 4, [k for k, _ in kernels]
  for name, body in kernels:
    lines = body.splitlines()
    # hand-written waits: the line after ;;#ASMSTART
    mine = {i + 1 for i, l in enumerate(lines) if 'ASMSTART' in l and i + 1 < len(lines) and 'vmcnt' in lines[i + 1]}
    assert len(mine) >= 6, (name, len(mine))
    # the loop's barriers are the hand-written ones (s_waitcnt lgkmcnt(0) + s_barrier in one asm block); the epilogue's
    # __syncthreads() come after the last of them
    barriers = [i for i, l in enumerate(lines) if re.search(r'\bs_barrier\b', l) and any('ASMSTART' in p for p in lines[max(0, i - 3):i])]
    dmas = [i for i, l in enumerate(lines) if 'global_load_lds_dwordx4' in l]
    assert dmas and len(barriers) >= 2, name
    lo, hi = dmas[0], barriers[-1]
    drained = [i for i in range(lo, hi) if re.search(r's_waitcnt.*vmcnt\(0\)', lines[i]) and i not in mine]
    assert not drained, (name, [lines[i].strip() for i in drained][:4], lo, hi)


def test_halo_convolution_kernels_do_not_spill_registers():
  """automl_amd/csrc/conv_halo.hip: 20 instantiations (seven input widths at stride 1, three at stride 2, x two column-tile layouts); the halo is loaded
  in passes of at most six chunks per thread because nine at once spilled vector registers at 96 channels (r06at).  No
  instantiation may spill or use scratch memory."""
  import os
  import re
  import subprocess
  import tempfile
  from automl_amd import build
  src = os.path.join(build.CSRC, 'conv_halo.hip')
  with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run([build.HIPCC] + build.FLAGS + ['-c', src, '-o', os.path.join(tmp, 'c.o'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  names = re.findall(r'Function Name: (\S+)', r.stderr)
  spills = [int(v) for v in re.findall(r'VGPRs Spill: (\d+)', r.stderr)]
  scratch = [int(v) for v in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', r.stderr)]
  assert len(names) == len(spills) == len(scratch) == 20, (len(names), len(spills), len(scratch))
  bad = [(n, s, c) for n, s, c in zip(names, spills, scratch) if s or c]
  assert not bad, bad
