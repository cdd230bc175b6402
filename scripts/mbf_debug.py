import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import _lib
from automl_amd._lib import ACT_SWISH, call, ptr
from tests import gpu_util as gu
from tests.test_gpu_mbconv_fused import _problem, _device_inputs, _bf
shape, ks = (1, 70, 67, 24, 144), (3, 2)
n, h, w, cin, cexp = shape
k, s = ks
x, isc, ish, wk, esc, esh, dww, e = _problem(shape, ks, False, ACT_SWISH)
er = _bf(e)
xd, wt, ldk, tv = _device_inputs(x, isc, ish, wk, cin, cexp)
oh, ow = (h + s - 1) // s, (w + s - 1) // s
escd, eshd, dwwd = gu.fdev(esc), gu.fdev(esh), gu.fdev(dww)
parts = torch.zeros(_lib.MAX_PARTS * 2 * cexp, dtype=torch.float32, device=gu.DEV)
npart = ctypes.c_int(0)
for trial in range(200):
  out = torch.full((n, oh, ow, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV)
  eout = torch.full((n, h, w, cexp), float('nan'), dtype=torch.bfloat16, device=gu.DEV)
  call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), ldk, cexp, ptr(escd), ptr(eshd), ACT_SWISH,
       ptr(eout), cexp, ptr(dwwd), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart), _lib.EDET_BF16, gu.stream())
  torch.cuda.synchronize()
  d = (eout.float().cpu() - er).abs()
  bad = (d > 0.1).nonzero()
  if len(bad):
    print('trial', trial, 'bad', len(bad), 'rows', sorted(set(bad[:, 1].tolist()))[:20], 'cols', sorted(set(bad[:, 2].tolist()))[:40],
          'ch', sorted(set(bad[:, 3].tolist()))[:60])
    print(bad[:10].tolist(), eout.float().cpu()[tuple(bad[0].tolist())], er[tuple(bad[0].tolist())])
    break
else:
  print('no failure in 200 trials')
