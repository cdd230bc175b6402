# Lab r06aa: k_big_gemm forward with PF register sets + coefficients requested with the chunk, against the round-5 kernel
# (automl_amd/libedet_hip_old.so = HEAD's pw_big.hip).  usage (gpurun): bash scripts/r06aa_big_pf.sh
mkdir -p gpurun_out; T=r06aa; export TMPDIR=/tmp; export EDET_SKIP_SLOW=1
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_effnetv2.py -m gpu -x -q -k "pw or conv or dense" 2>&1 | tail -5) > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log
L() { echo "== $*"; timeout 600 python scripts/kernel_lab.py "$@" 2>&1 | grep -v "^$" | tail -40; }
(
L --entry pw_fwd --layers mid --lib automl_amd/libedet_hip_old.so
L --entry pw_fwd --layers mid --ab EDET_BIG_PF=1,2,3
for sh in 8x192x192x384x384 8x96x96x384x384 8x48x48x384x384; do
  EDET_LAB_PLAIN=1 L --entry pw_fwd --shape $sh --lib automl_amd/libedet_hip_old.so
  EDET_LAB_PLAIN=1 L --entry pw_fwd --shape $sh --ab EDET_BIG_PF=1,2,3,4
done
for sh in 8x96x96x1344x224 8x48x48x2304x384 8x96x96x224x1344 8x192x192x480x80 8x384x384x288x48; do
  L --entry pw_fwd --shape $sh --lib automl_amd/libedet_hip_old.so
  L --entry pw_fwd --shape $sh --ab EDET_BIG_PF=1,2,3
done
) > gpurun_out/${T}_lab.log 2>&1
grep -E "^==|TOTAL|custom" gpurun_out/${T}_lab.log | cut -c1-160
