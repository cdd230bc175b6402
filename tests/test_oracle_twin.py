"""The torch oracle's ops equal an independent direct-loop numpy implementation on small shapes
(the oracle's own validation)."""
import numpy as np
import pytest
import torch

from oracle import direct_loops as dl
from oracle import efficientdet_oracle as orc


def nchw(x):
  return torch.from_numpy(x).permute(0, 3, 1, 2)


def nhwc(t):
  return t.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize('h,w', [(8, 8), (7, 9), (5, 4), (1, 1)])
@pytest.mark.parametrize('k,s', [(3, 1), (3, 2), (5, 1), (5, 2), (1, 1)])
def test_conv_and_depthwise(h, w, k, s):
  rng = np.random.default_rng(h * 100 + w * 10 + k + s)
  x = rng.standard_normal((2, h, w, 6)).astype(np.float32)
  wk = rng.standard_normal((k, k, 6, 5)).astype(np.float32)
  got = nhwc(orc.conv2d_same(nchw(x), torch.from_numpy(wk), s))
  np.testing.assert_allclose(got, dl.conv2d_same(x, wk, s), rtol=1e-4, atol=1e-5)
  wd = rng.standard_normal((k, k, 6)).astype(np.float32)
  got = nhwc(orc.depthwise_same(nchw(x), torch.from_numpy(wd.reshape(k, k, 6, 1)), s))
  np.testing.assert_allclose(got, dl.depthwise_same(x, wd, s), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('h,w', [(8, 8), (7, 9), (5, 4), (2, 3), (1, 1)])
def test_pool_and_resize(h, w):
  rng = np.random.default_rng(h * 10 + w)
  x = rng.standard_normal((2, h, w, 3)).astype(np.float32) - 2.0   # negative values: padding must not win
  np.testing.assert_array_equal(nhwc(orc.max_pool_same_3x3_s2(nchw(x))), dl.max_pool_3x3_s2_same(x))
  for oh, ow in ((2 * h, 2 * w), (2 * h - 1, 2 * w + 1), (h, w)):
    np.testing.assert_array_equal(nhwc(orc.resize_nearest(nchw(x), oh, ow)), dl.resize_nearest(x, oh, ow))


def test_batchnorm_training_and_moving_stats():
  rng = np.random.default_rng(1)
  x = (rng.standard_normal((3, 5, 4, 8)) * 2 + 1).astype(np.float32)
  g = rng.standard_normal(8).astype(np.float32)
  b = rng.standard_normal(8).astype(np.float32)
  o = orc.Oracle('efficientdet-d0', params={
      'bn/gamma': torch.from_numpy(g), 'bn/beta': torch.from_numpy(b),
      'bn/moving_mean': torch.zeros(8), 'bn/moving_variance': torch.ones(8)})
  got = nhwc(o.bn(nchw(x), 'bn', True))
  want, mean, var = dl.batch_norm_train(x, g, b)
  np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
  n = 3 * 5 * 4
  np.testing.assert_allclose(o.new_moving['bn/moving_mean'].numpy(), 0.01 * mean, rtol=1e-4, atol=1e-6)
  np.testing.assert_allclose(o.new_moving['bn/moving_variance'].numpy(), 0.99 + 0.01 * var * n / (n - 1), rtol=1e-4)


def test_swish_and_fusion():
  x = np.linspace(-4, 4, 9)
  np.testing.assert_allclose(orc.swish(torch.from_numpy(x)).numpy(), dl.swish(x), rtol=1e-6)
  np.testing.assert_allclose(dl.fast_attention([np.array([1.0, 3.0])] * 2, [1.0, 1.0]), [0.99995, 2.99985],
                             rtol=1e-6)


def test_losses_against_closed_forms():
  logits = torch.tensor([[0.3, -1.2, 2.0]])
  t = torch.tensor([[1.0, 0.0, 0.0]])
  got = orc.focal_loss(logits, t, 0.25, 1.5, 2.0).numpy()[0]
  p = 1 / (1 + np.exp(-logits.numpy()[0]))
  want = [0.25 * (1 - p[0])**1.5 * -np.log(p[0]) / 2, 0.75 * p[1]**1.5 * -np.log(1 - p[1]) / 2,
          0.75 * p[2]**1.5 * -np.log(1 - p[2]) / 2]
  np.testing.assert_allclose(got, want, rtol=1e-5)
  e = torch.tensor([0.05, -0.1, 0.3, -2.0])
  np.testing.assert_allclose(orc.huber(e, 0.1).numpy(), [0.00125, 0.005, 0.1 * 0.3 - 0.005, 0.1 * 2 - 0.005],
                             rtol=1e-6)


def test_same_padding_rule_is_stated_twice_and_agrees():
  """The TF 'SAME' geometry exists once in the product (automl_amd.utils.same_padding, used by the host code that sizes
  every stencil launch) and once, in another algebraic form, in the oracle (tf_same_pads).  Documented cases of the TF
  convolution guide plus an exhaustive sweep over the sizes / windows / strides the networks use."""
  from automl_amd import utils
  from oracle import efficientdet_oracle as orc
  # (size, window, stride) -> (before, after): the odd pixel goes after; no padding for 1x1; k < s never pads negative
  known = {(5, 3, 2): (1, 1), (4, 3, 2): (0, 1), (640, 3, 2): (0, 1), (7, 5, 2): (2, 2), (8, 5, 2): (1, 2),
           (13, 1, 1): (0, 0), (13, 3, 1): (1, 1), (13, 5, 1): (2, 2), (6, 1, 2): (0, 0), (5, 2, 2): (0, 1),
           (10, 3, 2): (0, 1), (9, 3, 2): (1, 1)}
  for (size, k, s), want in known.items():
    assert orc.tf_same_pads(size, k, s) == want, (size, k, s)
  for size in range(1, 130):
    for k in (1, 2, 3, 5, 7):
      for s in (1, 2, 3):
        out, before, after = utils.same_padding(size, k, s)
        assert (before, after) == orc.tf_same_pads(size, k, s), (size, k, s)
        assert out == -(-size // s) and (out - 1) * s + k <= size + before + after


@pytest.mark.parametrize('h,w', [(8, 8), (7, 9), (5, 4)])
@pytest.mark.parametrize('k,s', [(3, 1), (3, 2), (5, 1), (5, 2)])
def test_conv_against_scipy_as_a_third_witness(h, w, k, s):
  """A third, library-independent statement of the 'SAME' convolution: scipy.signal.correlate2d over an explicitly
  zero-padded image, sub-sampled at the stride -- padding written here from the TF rule (extra pixel after), so the
  torch oracle, the direct loops and this agree only if all three place the window the same way."""
  from scipy import signal
  rng = np.random.default_rng(1000 + h * 100 + w * 10 + k + s)
  x = rng.standard_normal((1, h, w, 3)).astype(np.float64)
  wk = rng.standard_normal((k, k, 3, 2)).astype(np.float64)
  oh, ow = -(-h // s), -(-w // s)
  ph, pw_ = max((oh - 1) * s + k - h, 0), max((ow - 1) * s + k - w, 0)
  xp = np.pad(x[0], ((ph // 2, ph - ph // 2), (pw_ // 2, pw_ - pw_ // 2), (0, 0)))
  want = np.zeros((oh, ow, 2))
  for co in range(2):
    for ci in range(3):
      full = signal.correlate2d(xp[:, :, ci], wk[:, :, ci, co], mode='valid')
      want[:, :, co] += full[::s, ::s][:oh, :ow]
  got = nhwc(orc.conv2d_same(nchw(x.astype(np.float32)), torch.from_numpy(wk.astype(np.float32)), s))[0]
  np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(dl.conv2d_same(x.astype(np.float32), wk.astype(np.float32), s)[0], want, rtol=1e-4, atol=1e-5)
