# r03d: lab sweep of launch heuristics (in-process A/B through the per-call environment switches), D7x B=8 diagnostics.
mkdir -p gpurun_out
T=${1:-r03d}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
($L --entry pw_bwd --layers big --ab EDET_PWS_FUSED_GRID=1024,512,256 2>&1 | tail -20) > gpurun_out/${T}_lab_fgrid.log
($L --entry pw_bwd --layers big --ab EDET_PW_NOY=0,1 2>&1 | tail -12) > gpurun_out/${T}_lab_noy.log
($L --entry pw_fwd --layers all --ab EDET_PWS_SPW=2,4,8,16 2>&1 | tail -110) > gpurun_out/${T}_lab_spw_fwd.log
($L --entry pw_bwd --layers all --ab EDET_PWS_SPW=2,4,8 2>&1 | tail -80) > gpurun_out/${T}_lab_spw_bwd.log
($L --entry pw_bwd --layers all --ab EDET_PWS_WG_TARGET=2048,1024,512 2>&1 | tail -80) > gpurun_out/${T}_lab_pwswg.log
($L --entry dw_fwd --layers all --ab EDET_DWM_P=4096,2048,8192 2>&1 | tail -50) > gpurun_out/${T}_lab_dwp_fwd.log
($L --entry dw_bwd --layers all --ab EDET_DWM_P=4096,2048,8192 2>&1 | tail -50) > gpurun_out/${T}_lab_dwp_bwd.log
(timeout 900 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "batch8_train" tests/test_gpu_kernels.py -k "squeeze_excite or test_pw_bwd" 2>&1 | grep -v "^$" | cut -c1-1600 | tail -30) > gpurun_out/${T}_side.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 600 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x.txt 2>&1 | tail -1 | cut -c1-300) > gpurun_out/${T}_bench_d7x.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -12 gpurun_out/${T}_side.log | cut -c1-1200; cut -c1-330 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_d7x.log
