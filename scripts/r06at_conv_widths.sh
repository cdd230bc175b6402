# r06at: the halo-tile convolution at the other Fused-MBConv widths of the EfficientNetV2 family: parity, then per-shape A/B
mkdir -p gpurun_out
export TMPDIR=/tmp EDET_SKIP_SLOW=1
(timeout 900 python -m pytest tests/test_effnetv2.py -m gpu -x -q -p no:cacheprovider -k "conv_fwd" 2>&1 | tail -8) > gpurun_out/r06at_pytest.log; tail -4 gpurun_out/r06at_pytest.log | cut -c1-400
python scripts/bench_conv.py --env EDET_CONV_HALO_WIDTHS=all --ab EDET_CONV_HALO=0,1 --shapes 128x112x112x32x16,128x112x112x16x16,128x56x56x32x128,128x112x112x32x32,128x56x56x16x64,128x28x28x80x320,128x28x28x96x384,128x112x112x24x24,128x56x56x48x192,128x28x28x64x256,128x56x56x64x256,128x14x14x96x384 2>&1 | grep "^conv3x3" | tee gpurun_out/r06at_conv.txt
