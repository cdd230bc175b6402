# r06ak..: LDS-DMA wide pointwise forward (pw_glds.hip) -- parity + bit equality, per-layer A/B, step A/B on the three configs
T=${1:-r06ak}
mkdir -p gpurun_out
export TMPDIR=/tmp
LAB_PYTEST="tests/test_gpu_kernels.py" LAB_PYTEST_K="test_pw_fwd" \
LAB_KERNEL="pw_fwd|mid|EDET_PW_GLDS=0,1,2" \
LAB_STEPS=10 LAB_BENCH="${LAB_BENCH-g0:EDET_PW_GLDS=0;g2:EDET_PW_GLDS=2;g0b:EDET_PW_GLDS=0;g2b:EDET_PW_GLDS=2}" bash scripts/gpu_lab.sh $T
echo "== plain views"
EDET_LAB_PLAIN=1 timeout 300 python scripts/kernel_lab.py --entry pw_fwd --layers mid --ab EDET_PW_GLDS=0,1,2 2>&1 | grep -v "fpn_\|rs_\|cls_\|box_" | tail -40 | cut -c1-150
echo "== d7x / v2s shapes"
for sh in 8x192x192x384x384 8x96x96x384x384 8x96x96x1344x224 8x96x96x224x1344 8x48x48x2304x384 8x48x48x384x2304 8x48x48x3840x640 8x384x384x288x48 256x14x14x960x160 256x14x14x160x960 256x56x56x192x48; do
  timeout 120 python scripts/kernel_lab.py --entry pw_fwd --shape $sh --ab EDET_PW_GLDS=0,1,2 2>&1 | grep "^pw_fwd" | cut -c1-150
  EDET_LAB_PLAIN=1 timeout 120 python scripts/kernel_lab.py --entry pw_fwd --shape $sh --ab EDET_PW_GLDS=0,1,2 2>&1 | grep "^pw_fwd" | sed 's/^pw_fwd/plain /' | cut -c1-150
done
if [ -n "$LAB_SIDE" ]; then
for g in 0 $LAB_SIDE; do
  echo "== d7x EDET_PW_GLDS=$g"; (EDET_PW_GLDS=$g timeout 600 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | grep "^{" | tail -1 | cut -c1-200) | tee gpurun_out/${T}_d7x_g$g.json
  echo "== v2s EDET_PW_GLDS=$g"; (EDET_PW_GLDS=$g timeout 300 python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_v2s_g${g}_launches.txt 2>&1 | grep "^{" | tail -1 | cut -c1-200) | tee gpurun_out/${T}_v2s_g$g.json
done
fi
