import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from automl_amd import _lib
from automl_amd._lib import ACT_NONE, ACT_SWISH, TView, call, ptr
dev = 'cuda:0'
st = torch.cuda.current_stream().cuda_stream
bf = _lib.EDET_BF16
for (n, hw, cin, cexp, k, s) in ((128, 320, 16, 96, 3, 2), (128, 160, 24, 144, 3, 1), (128, 160, 24, 144, 5, 2)):
  g = torch.Generator(device=dev).manual_seed(1)
  x1 = torch.randn((2, hw, hw, cin), generator=g, device=dev).to(torch.bfloat16)
  x = x1.repeat(n // 2, 1, 1, 1).contiguous()
  wk = (torch.randn((cin, cexp), generator=g, device=dev) / np.sqrt(cin)).float()
  wt = torch.zeros(cexp, cin, dtype=torch.bfloat16, device=dev)
  call('edet_cast_matrix', ptr(wk), ptr(wt), cin, cexp, cin, 1, bf, st)
  esc = (1 + 0.1 * torch.randn(cexp, generator=g, device=dev)).float()
  esh = (0.1 * torch.randn(cexp, generator=g, device=dev)).float()
  dww = (torch.randn((k, k, cexp), generator=g, device=dev) / k).float()
  oh = (hw + s - 1) // s
  outs = {}
  for nn, xx in ((2, x1), (n, x)):
    e = torch.full((nn, hw, hw, cexp), float('nan'), dtype=torch.bfloat16, device=dev)
    out = torch.full((nn, oh, oh, cexp), float('nan'), dtype=torch.bfloat16, device=dev)
    parts = torch.zeros(_lib.MAX_PARTS * 2 * cexp, dtype=torch.float32, device=dev)
    npart = ctypes.c_int(0)
    tv = TView(ptr(xx), None, None, None, ACT_NONE, nn, hw, hw, cin, cin)
    call('edet_mbconv_expand_dw_fwd', ctypes.byref(tv), ptr(wt), cin, cexp, ptr(esc), ptr(esh), ACT_SWISH,
         ptr(e), cexp, ptr(dww), k, s, ptr(out), cexp, ptr(parts), ctypes.byref(npart), bf, st)
    torch.cuda.synchronize()
    outs[nn] = (e, out, npart.value)
    for nm, t in (('E', e), ('out', out)):
      bad = (~torch.isfinite(t.float())).nonzero()
      print(nn, hw, cexp, k, s, nm, 'P', npart.value, 'nonfinite', len(bad), bad[:6].tolist(),
            'imgs', sorted(set(bad[:, 0].tolist()))[:10], 'rows', sorted(set(bad[:, 1].tolist()))[:10],
            'cols', sorted(set(bad[:, 2].tolist()))[:10], flush=True)
  e2, o2, _ = outs[2]
  e128, o128, _ = outs[n]
  for kk in (0, n // 4, n // 2 - 1):
    print('  pair', kk, 'E equal', bool(torch.equal(e128[2 * kk:2 * kk + 2], e2)), 'out equal', bool(torch.equal(o128[2 * kk:2 * kk + 2], o2)))
  d = (e128[0:2].float() != e2.float()).nonzero()
  print('  mismatches in pair 0:', len(d), 'rows', sorted(set(d[:, 1].tolist()))[:12], 'cols', sorted(set(d[:, 2].tolist()))[:24], 'ch', sorted(set(d[:, 3].tolist()))[:32])
  if len(d):
    i = tuple(d[0].tolist())
    print('   first', i, float(e128[0:2].float()[i]), float(e2.float()[i]))
  if len(d):
    r0, c0 = int(d[0][1]), int(d[0][2])
    sel = d[(d[:, 0] == d[0][0]) & (d[:, 1] == r0)]
    for cc in sorted(set(sel[:, 2].tolist()))[:6]:
      print('   row', r0, 'col', cc, 'bad ch', sel[sel[:, 2] == cc][:, 3].tolist())
