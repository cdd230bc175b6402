# Lab r06ae: k_big_gemm forward with 64 / 32-row tiles for the layers with few rows (EDET_BIG_BM), parity first
mkdir -p gpurun_out; T=r06ae; export TMPDIR=/tmp; export EDET_SKIP_SLOW=1
for bm in 32 64; do
  (EDET_BIG_BM=$bm EDET_PW_IMPL=big timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pw_fwd or pw_forward or pointwise_fwd" 2>&1 | grep -E "passed|failed|error" | tail -2) > gpurun_out/${T}_pytest_bm$bm.log; echo "bm=$bm big: $(cat gpurun_out/${T}_pytest_bm$bm.log)"
  (EDET_BIG_BM=$bm timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_effnetv2.py tests/test_gpu_side_configs.py -m gpu -x -q -k "pw or v2 or d7x" 2>&1 | grep -E "passed|failed|error" | tail -2) > gpurun_out/${T}_pytest2_bm$bm.log; echo "bm=$bm auto: $(cat gpurun_out/${T}_pytest2_bm$bm.log)"
done
(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -x -q -k 'adam or test_optimizer' 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -30 | cut -c1-600) > gpurun_out/${T}_adam.log; tail -12 gpurun_out/${T}_adam.log
L() { echo "== $*"; timeout 600 python scripts/kernel_lab.py "$@" 2>&1 | grep -v "^$" | tail -40; }
(
for sh in 256x7x7x1536x256 256x7x7x256x1536 256x14x14x960x160 256x14x14x160x960 256x14x14x512x128 256x14x14x128x512 256x28x28x256x64 \
          8x48x48x2304x384 8x48x48x384x2304 8x48x48x3840x640 8x48x48x640x3840 8x96x96x1344x224 8x96x96x960x160 8x96x96x224x1344 \
          128x20x20x1152x192 128x20x20x192x1152 128x20x20x1152x320 128x20x20x672x192 128x40x40x672x112; do
  L --entry pw_fwd --shape $sh --ab EDET_BIG_BM=128,64,32
done
for sh in 8x48x48x384x384 8x24x24x384x384 128x20x20x64x64 128x10x10x64x64; do
  EDET_LAB_PLAIN=1 L --entry pw_fwd --shape $sh --ab EDET_BIG_BM=128,64,32
done
) > gpurun_out/${T}_lab.log 2>&1
grep -E "custom" gpurun_out/${T}_lab.log | cut -c1-150
