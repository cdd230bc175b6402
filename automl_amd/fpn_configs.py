"""BiFPN node graph (reference efficientdet/tf2/fpn_configs.py:24-72,166-176)."""
from automl_amd import hparams_config


def bifpn_config(min_level, max_level, weight_method):
  """Top-down then bottom-up node list for levels min_level..max_level.

  Node ids: 0..L-1 are the input levels; every new node takes the next id.
  Top-down node at level i fuses [last(i), last(i+1)]; bottom-up node at level
  i fuses [all ids of level i] + [last(i-1)].
  """
  p = hparams_config.Config()
  p.weight_method = weight_method or 'fastattn'
  num_levels = max_level - min_level + 1
  ids = {min_level + i: [i] for i in range(num_levels)}
  next_id = num_levels
  nodes = []
  for lvl in range(max_level - 1, min_level - 1, -1):
    nodes.append({'feat_level': lvl,
                  'inputs_offsets': [ids[lvl][-1], ids[lvl + 1][-1]]})
    ids[lvl].append(next_id)
    next_id += 1
  for lvl in range(min_level + 1, max_level + 1):
    nodes.append({'feat_level': lvl,
                  'inputs_offsets': list(ids[lvl]) + [ids[lvl - 1][-1]]})
    ids[lvl].append(next_id)
    next_id += 1
  p.nodes = nodes
  return p


def get_fpn_config(fpn_name, min_level, max_level, weight_method):
  if not fpn_name:
    fpn_name = 'bifpn'
  if fpn_name in ('bifpn', 'bifpn_dyn'):
    return bifpn_config(min_level, max_level, weight_method)
  raise ValueError('fpn_name %r is out of scope (only bifpn is built)' % fpn_name)
