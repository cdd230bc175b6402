import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from automl_amd import hparams_config, train_lib
sys.path.insert(0, '/root/repo/tests')
from tests.test_gpu_network import perturbed_params, make_labels
model, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
config = hparams_config.get_efficientdet_config(model)
vals = perturbed_params(config, 11)
rng = np.random.default_rng(97)
images = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
labels = make_labels(config, batch, size, 101)
runs = []
for _ in range(2):
  net = train_lib.EfficientDetNetTrain(config=config, dtype='bf16', params=vals, seed=5)
  eng = net._ensure_engine(batch, size, size)
  eng.refresh_drop_masks()
  eng.forward(net._to_device_images(images, eng), training=True)
  eng.loss_backward(net._labels_to_device(labels, eng))
  torch.cuda.synchronize()
  runs.append((eng.grads_flat.clone(), {k: (m.clone()) for k, (m, p) in eng.drop_masks.items()}))
g0, g1 = runs[0][0], runs[1][0]
for k in runs[0][1]:
  if not torch.equal(runs[0][1][k], runs[1][1][k]): print('MASK differs', k)
bad = []
for name in eng.seg_names:
  off, n, shape, tr = eng.offsets[name]
  if not torch.equal(g0[off:off+n], g1[off:off+n]):
    d = (g0[off:off+n] - g1[off:off+n]).abs().max().item()
    bad.append((name, tuple(shape), d, g0[off:off+n].abs().max().item()))
print(len(bad), 'tensors differ')
for b in bad[-40:]: print(b)
