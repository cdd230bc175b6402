# r03m: epilogue lane mapping by chunk width, scalar-base addressing, transpose reads for the weight-gradient fragments.
mkdir -p gpurun_out
T=${1:-r03m}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "test_pw_bwd" 2>&1 | cut -c1-2500 | tail -30) > gpurun_out/${T}_kern.log
(EDET_PWS_TR=1 timeout 900 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "test_pw_bwd and bf16 and not data and not weight" 2>&1 | cut -c1-1500 | tail -15) > gpurun_out/${T}_kern_tr.log
($L --entry pw_bwd --layers all --ab EDET_PWS_TR=0,1 2>&1 | tail -60) > gpurun_out/${T}_lab_tr.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -4 gpurun_out/${T}_kern.log | cut -c1-800; tail -6 gpurun_out/${T}_kern_tr.log | cut -c1-800; grep -E "b0_project|b1_|b2_|b3_project|fpn_80|TOTAL" gpurun_out/${T}_lab_tr.log | cut -c1-140; cut -c150-330 gpurun_out/${T}_bench_b128.log
