"""Generates tests/golden/reference_v2_tables.json by importing the REFERENCE's own efficientnetv2 modules.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_v2.py
efficientnetv2/hparams.py imports TensorFlow only for yaml file I/O and effnetv2_configs.py imports nothing
else, so a stub module named `tensorflow` is enough to import and execute both unmodified.  For every model
name the reference knows, the table holds the merged model config (hparams.base_config.model overridden by
effnetv2_configs.get_model_config(name).model, what EffNetV2Model.__init__ computes,
effnetv2_model.py:549-556) with the decoded block list.
"""
import copy
import json
import os
import sys
import types

REF = '/root/reference/efficientnetv2'


def main():
  tf = types.ModuleType('tensorflow')
  tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace(GFile=open))
  sys.modules['tensorflow'] = tf
  sys.path.insert(0, REF)
  import effnetv2_configs as ref_cfg      # noqa: the reference module
  import hparams as ref_hp                # noqa
  out = {}
  names = list(ref_cfg.efficientnetv1_params) + list(ref_cfg.efficientnetv2_params)
  for name in names:
    cfg = copy.deepcopy(ref_hp.base_config)
    cfg.override(ref_cfg.get_model_config(name))
    m = cfg.model.as_dict()
    m['blocks_args'] = [b.as_dict() for b in cfg.model.blocks_args]
    out[name] = {'model': m, 'train_isize': cfg.train.isize, 'eval_isize': cfg.eval.isize}
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, 'reference_v2_tables.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print('wrote', len(out), 'models')


if __name__ == '__main__':
  main()
