# r03f: the r03d / r03e choices re-measured with the rotated lab (no first-variant bias), bench with the new dispatch.
mkdir -p gpurun_out
T=${1:-r03f}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
($L --entry dw_bwd --layers all --ab EDET_DWM_P=unset,2048,4096,8192 2>&1 | tail -70) > gpurun_out/${T}_lab_dwp_bwd.log
($L --entry dw_fwd --layers all --ab EDET_DWM_P=unset,2048,4096,8192 2>&1 | tail -70) > gpurun_out/${T}_lab_dwp_fwd.log
($L --entry pw_bwd --layers big --ab EDET_PWS_FUSED_GRID=256,512,1024 2>&1 | tail -20) > gpurun_out/${T}_lab_fgrid.log
($L --entry pw_bwd --layers big --ab EDET_PW_NOY=1,0 2>&1 | tail -12) > gpurun_out/${T}_lab_noy.log
($L --entry pw_fwd --layers all --ab EDET_PWS_SPW=4,2,8 2>&1 | tail -80) > gpurun_out/${T}_lab_spw_fwd.log
($L --entry pw_bwd --layers all --ab EDET_PWS_SPW=4,2,8 2>&1 | tail -80) > gpurun_out/${T}_lab_spw_bwd.log
($L --entry pw_bwd_weight --layers all --ab EDET_WGRAD_WGS=unset,512,1024,2048 2>&1 | tail -100) > gpurun_out/${T}_lab_wgs.log
($L --entry pw_bwd --layers all --ab EDET_BIG_TPW=1,2 2>&1 | tail -60) > gpurun_out/${T}_lab_tpw.log
(timeout 600 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "batch8_train" 2>&1 | grep -v "^$" | cut -c1-1800 | tail -12) > gpurun_out/${T}_side.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 300 python scripts/bench_v2s.py --steps 20 2>&1 | tail -1 | cut -c1-400) > gpurun_out/${T}_bench_v2s.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -6 gpurun_out/${T}_side.log | cut -c1-1500; cut -c1-330 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_v2s.log
