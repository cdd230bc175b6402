# r06au: the halo-tile convolution at stride 2: parity, per-shape A/B, V2-S forward
mkdir -p gpurun_out
export TMPDIR=/tmp EDET_SKIP_SLOW=1
(timeout 900 python -m pytest tests/test_effnetv2.py -m gpu -x -q -p no:cacheprovider -k "conv_fwd" 2>&1 | tail -8) > gpurun_out/r06au_pytest.log; tail -4 gpurun_out/r06au_pytest.log | cut -c1-600
python scripts/bench_conv.py --stride 2 --ab EDET_CONV_HALO=0,1 --shapes 256x112x112x24x96,256x56x56x48x192,128x112x112x16x64,128x56x56x32x128,128x112x112x32x128 2>&1 | grep "^conv3x3" | tee gpurun_out/r06au_conv.txt
for h in 0 1 0 1; do
  echo "== v2s EDET_CONV_HALO=$h"; (EDET_CONV_HALO=$h timeout 300 python scripts/bench_v2s.py --steps 10 2>&1 | grep "^{" | tail -1 | cut -c1-200)
done
