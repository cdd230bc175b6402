# r03s: class A3 of the one-pass pointwise backward (40 -> 240).
mkdir -p gpurun_out
T=${1:-r03s}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "test_pw_bwd and (40-240 or 48-256 or 16-96)" 2>&1 | cut -c1-2500 | tail -12) > gpurun_out/${T}_kern.log
($L --entry pw_bwd --layers b4_expand --ab EDET_PWS_FUSED_A3=0,1 2>&1 | tail -6) > gpurun_out/${T}_lab_a3.log
tail -5 gpurun_out/${T}_kern.log | cut -c1-1200; cat gpurun_out/${T}_lab_a3.log | grep -v amdgpu | cut -c1-140
