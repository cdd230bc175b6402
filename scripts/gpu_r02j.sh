# round 2, call j: depthwise lab 2 (fused PF 8, tile height 40/64/80, deeper forward FIFO, stride-2 gradient kernels with
# deeper FIFOs at 2 waves/SIMD), the tests recalibrated after r02i, bench with the r02i winners hardwired
mkdir -p gpurun_out
T=r02j
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_checkpoint.py -m gpu -q 2>&1 | grep -v "^$" | cut -c1-2000 | tail -80) > gpurun_out/${T}_ckpt.log
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dw or other_activations" 2>&1 | grep -v "^$" | cut -c1-2000 | tail -30) > gpurun_out/${T}_kern.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 200 python scripts/kernel_lab.py --entry dw_bwd --layers all --ab EDET_DWB_PF=0,8 2>&1 | tail -40) > gpurun_out/${T}_lab_pf.log
(timeout 200 python scripts/kernel_lab.py --entry dw_bwd --layers all --ab EDET_DW_TY=40,64,80 2>&1 | tail -50) > gpurun_out/${T}_lab_ty_bwd.log
(timeout 200 python scripts/kernel_lab.py --entry dw_fwd --layers all --ab EDET_DW_TY=40,64,80 2>&1 | tail -50) > gpurun_out/${T}_lab_ty_fwd.log
(timeout 200 python scripts/kernel_lab.py --entry dw_fwd --layers all --ab EDET_DWF_PF=0,1 2>&1 | tail -40) > gpurun_out/${T}_lab_fpf.log
(timeout 200 python scripts/kernel_lab.py --entry dw_bwd --layers all --ab EDET_DWS2_PF=0,1 2>&1 | tail -40) > gpurun_out/${T}_lab_s2.log
tail -4 gpurun_out/${T}_ckpt.log | cut -c1-400; tail -3 gpurun_out/${T}_kern.log | cut -c1-300; cut -c1-330 gpurun_out/${T}_bench_b128.log
for f in pf ty_bwd ty_fwd fpf s2; do grep TOTAL gpurun_out/${T}_lab_$f.log; done
