# r03o: streaming forward with unconditional load passes (EXACT instantiations); cost of the statistics epilogue.
mkdir -p gpurun_out
T=${1:-r03o}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "test_pw_fwd" 2>&1 | cut -c1-2500 | tail -8) > gpurun_out/${T}_kern.log
($L --entry pw_fwd --layers all --ab EDET_PWS_FWD_EXACT=0,1 2>&1 | tail -60) > gpurun_out/${T}_lab_exact.log
($L --entry pw_fwd --layers all --ab EDET_LAB_NOSTATS=0,1 2>&1 | tail -60) > gpurun_out/${T}_lab_nostats.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -3 gpurun_out/${T}_kern.log | cut -c1-800; cat gpurun_out/${T}_lab_exact.log gpurun_out/${T}_lab_nostats.log | grep -v amdgpu | cut -c1-140; cut -c150-330 gpurun_out/${T}_bench_b128.log
