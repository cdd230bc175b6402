# r03i: occupancy-derived forward cap (bench), depthwise persistent workgroups as multiples of the resident count (lab).
mkdir -p gpurun_out
T=${1:-r03i}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 600 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "detection_loss or test_pw_fwd" 2>&1 | cut -c1-1500 | tail -4) > gpurun_out/${T}_kern.log
($L --entry pw_fwd --layers all --ab EDET_PWS_FWD_CAP=unset,1024,768,512 2>&1 | tail -110) > gpurun_out/${T}_lab_fwdcap.log
($L --entry dw_fwd --layers all --ab EDET_DWM_ROUNDS=unset,1,1.5,2,3,4 2>&1 | tail -130) > gpurun_out/${T}_lab_dwr_fwd.log
($L --entry dw_bwd --layers all --ab EDET_DWM_ROUNDS=unset,1,1.5,2,3,4 2>&1 | tail -130) > gpurun_out/${T}_lab_dwr_bwd.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 300 python scripts/bench_v2s.py --steps 20 2>&1 | tail -1 | cut -c1-400) > gpurun_out/${T}_bench_v2s.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -3 gpurun_out/${T}_kern.log | cut -c1-600; cut -c1-330 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_v2s.log
