# r03n: one-pass pointwise backward, project shapes only in class B, transpose reads unconditional; G / in-flight sweeps.
mkdir -p gpurun_out
T=${1:-r03n}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "test_pw_bwd" 2>&1 | cut -c1-2500 | tail -30) > gpurun_out/${T}_kern.log
($L --entry pw_bwd --layers big --ab EDET_PWS_FUSED_G=1,2,4 2>&1 | tail -30) > gpurun_out/${T}_lab_g.log
($L --entry pw_bwd --layers big --ab EDET_PWS_FUSED_INFLIGHT=16384,24576,32768,49152 2>&1 | tail -30) > gpurun_out/${T}_lab_inflight.log
($L --entry pw_bwd --layers big --ab EDET_PWS_FUSED_GRID=256,512 2>&1 | tail -30) > gpurun_out/${T}_lab_grid.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -4 gpurun_out/${T}_kern.log | cut -c1-800; cat gpurun_out/${T}_lab_g.log gpurun_out/${T}_lab_inflight.log gpurun_out/${T}_lab_grid.log | grep -v amdgpu | cut -c1-140; cut -c150-330 gpurun_out/${T}_bench_b128.log
