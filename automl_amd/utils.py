"""Shape helpers of the detection path (reference efficientdet/utils.py:484-526)."""


def parse_image_size(image_size):
  """int | 'WxH' string | (H, W) tuple -> (height, width)."""
  if isinstance(image_size, int):
    return (image_size, image_size)
  if isinstance(image_size, str):
    width, height = image_size.lower().split('x')
    return (int(height), int(width))
  if isinstance(image_size, (tuple, list)):
    return tuple(image_size)
  raise ValueError('image_size must be an int, WxH string, or (height, width)'
                   'tuple. Was %r' % (image_size,))


def get_feat_sizes(image_size, max_level):
  """Feature (height, width) per level 0..max_level: s_{l+1} = (s_l - 1)//2 + 1."""
  h, w = parse_image_size(image_size)
  sizes = [{'height': h, 'width': w}]
  for _ in range(max_level):
    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    sizes.append({'height': h, 'width': w})
  return sizes


def same_padding(in_size, kernel, stride):
  """TensorFlow 'SAME' geometry: (out_size, pad_before, pad_after).

  out = ceil(in/stride); total = max((out-1)*stride + kernel - in, 0);
  before = total//2 (the extra pixel goes after: asymmetric for stride 2 on
  even inputs).  Used by every stencil kernel and by the oracle.
  """
  out = -(-in_size // stride)
  total = max((out - 1) * stride + kernel - in_size, 0)
  return out, total // 2, total - total // 2
