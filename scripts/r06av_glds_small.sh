# r06av: the LDS-DMA forward kernel on small maps (up to four images per row tile): equality test, per-shape A/B, V2-S forward
mkdir -p gpurun_out
export TMPDIR=/tmp EDET_SKIP_SLOW=1
(timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "test_pw_fwd" 2>&1 | tail -8) > gpurun_out/r06av_pytest.log; tail -3 gpurun_out/r06av_pytest.log | cut -c1-600
for sh in 256x7x7x1536x256 256x7x7x960x256 256x14x14x960x160; do
  timeout 120 python scripts/kernel_lab.py --entry pw_fwd --shape $sh --ab EDET_PW_GLDS=0,1 2>&1 | grep "^pw_fwd" | cut -c1-150
done
for g in 0 1 0 1; do
  echo "== v2s EDET_PW_GLDS=$g"; (EDET_PW_GLDS=$g timeout 300 python scripts/bench_v2s.py --steps 10 2>&1 | grep "^{" | tail -1 | cut -c1-200)
done
