# Lab r06ac: row tiles per workgroup of the tiled GEMM (EDET_BIG_TPW) on the D7x forward shapes and the D0 mid layers
mkdir -p gpurun_out; T=r06ac; export TMPDIR=/tmp
L() { echo "== $*"; timeout 600 python scripts/kernel_lab.py "$@" 2>&1 | grep -v "^$" | tail -40; }
(
for sh in 8x192x192x384x384 8x96x96x384x384; do
  EDET_LAB_PLAIN=1 L --entry pw_fwd --shape $sh --ab EDET_BIG_TPW=1,2,3,4,6,9
done
for sh in 8x96x96x1344x224 8x48x48x2304x384 8x96x96x224x1344 8x192x192x480x80 8x384x384x288x48 8x48x48x384x2304; do
  L --entry pw_fwd --shape $sh --ab EDET_BIG_TPW=1,2,3,4,6
done
L --entry pw_fwd --layers mid --ab EDET_BIG_TPW=1,2,3,4
L --entry pw_bwd_data --shape 8x96x96x1344x224 --ab EDET_BIG_TPW=1,2,3,4
) > gpurun_out/${T}_lab.log 2>&1
grep -E "^==|TOTAL|custom" gpurun_out/${T}_lab.log | cut -c1-160
