# r03p: depthwise forward / fused backward with every load of a row step issued unconditionally (precise FIFO waits).
mkdir -p gpurun_out
T=${1:-r03p}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "dw" 2>&1 | cut -c1-2500 | tail -8) > gpurun_out/${T}_kern.log
($L --entry dw_fwd --layers all 2>&1 | tail -20) > gpurun_out/${T}_lab_dw_fwd.log
($L --entry dw_bwd --layers all 2>&1 | tail -20) > gpurun_out/${T}_lab_dw_bwd.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -3 gpurun_out/${T}_kern.log | cut -c1-800; cat gpurun_out/${T}_lab_dw_fwd.log gpurun_out/${T}_lab_dw_bwd.log | grep -v amdgpu | cut -c1-140; cut -c150-330 gpurun_out/${T}_bench_b128.log
