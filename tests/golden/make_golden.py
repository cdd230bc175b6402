"""Generates tests/golden/reference_tables.json by importing the REFERENCE's own pure-Python modules.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference's hparams_config.py and tf2/fpn_configs.py import TensorFlow solely for yaml file I/O
(hparams_config.py:113,119), so a stub module named `tensorflow` is enough to import and execute
them unmodified.  The arithmetic modules (Keras layers) cannot be imported without TensorFlow.
"""
import json
import os
import sys
import types

REF = '/root/reference/efficientdet'


def main():
  tf = types.ModuleType('tensorflow')
  tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace(GFile=open))
  sys.modules['tensorflow'] = tf
  sys.path.insert(0, REF)
  sys.path.insert(0, os.path.join(REF, 'tf2'))
  import hparams_config as ref_hp          # noqa: the reference module
  import fpn_configs as ref_fpn            # noqa
  out = {'models': {}, 'bifpn': {}, 'override_cases': []}
  names = list(ref_hp.efficientdet_model_param_dict) + list(ref_hp.efficientdet_lite_param_dict)
  for name in names:
    out['models'][name] = ref_hp.get_efficientdet_config(name).as_dict()
  for lo, hi in ((3, 7), (2, 7), (3, 8), (3, 5)):
    for wm in (None, 'sum', 'attn'):
      cfg = ref_fpn.bifpn_config(lo, hi, wm)
      out['bifpn']['%d_%d_%s' % (lo, hi, wm)] = {'weight_method': cfg.weight_method,
                                                 'nodes': [dict(n) for n in cfg.nodes]}
  for s in ('image_size=640,mixed_precision=true', 'nms_configs.method=hard,num_classes=20',
            'aspect_ratios=1.0*2.0*0.5,image_size=1920x1280', 'anchor_scale=3.5,,,heads=a*b,'):
    c = ref_hp.get_efficientdet_config('efficientdet-d0')
    c.override(s)
    out['override_cases'].append({'str': s, 'result': c.as_dict()})
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, 'reference_tables.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print('wrote', len(out['models']), 'models,', len(out['bifpn']), 'bifpn graphs')


if __name__ == '__main__':
  main()
