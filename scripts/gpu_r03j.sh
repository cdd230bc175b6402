# r03j: one-pass pointwise backward widened to the project / 64->64 layers (several tiles per step, sums in registers).
mkdir -p gpurun_out
T=${1:-r03j}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "test_pw_bwd" 2>&1 | cut -c1-2500 | tail -30) > gpurun_out/${T}_kern.log
($L --entry pw_bwd --layers all --ab EDET_PWS_FUSED_WIDE=0,1 2>&1 | tail -60) > gpurun_out/${T}_lab_wide.log
($L --entry pw_bwd --layers all --ab EDET_PWS_FUSED_G=1,2,4 2>&1 | tail -90) > gpurun_out/${T}_lab_g.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -6 gpurun_out/${T}_kern.log | cut -c1-800; cat gpurun_out/${T}_lab_wide.log | cut -c1-140; grep TOTAL gpurun_out/${T}_lab_g.log; cut -c1-330 gpurun_out/${T}_bench_b128.log
