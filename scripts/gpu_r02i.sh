# round 2, call i: OACT-templated pointwise kernels (bench must be back at r02e), recalibrated tests, checkpoint test,
# lab: fused depthwise backward prefetch depth and tile height
mkdir -p gpurun_out
T=r02i
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_checkpoint.py "tests/test_gpu_network.py::test_train_step_matches_oracle_fp32" tests/test_gpu_bench_shapes.py::test_d0_640_batch2_bf16_train_step_layer_by_layer tests/test_gpu_kernels.py::test_tuned_kernels_with_the_other_activations -m gpu -q 2>&1 | grep -v "^$" | cut -c1-3000 | tail -120) > gpurun_out/${T}_tests.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 300 python scripts/kernel_lab.py --entry dw_bwd --layers all --ab EDET_DWB_PF=0,6 2>&1 | tail -40) > gpurun_out/${T}_lab_pf.log
(timeout 300 python scripts/kernel_lab.py --entry dw_bwd --layers all --ab EDET_DW_TY=0,40,20 2>&1 | tail -50) > gpurun_out/${T}_lab_ty_bwd.log
(timeout 300 python scripts/kernel_lab.py --entry dw_fwd --layers all --ab EDET_DW_TY=0,40,20 2>&1 | tail -50) > gpurun_out/${T}_lab_ty_fwd.log
tail -15 gpurun_out/${T}_tests.log | cut -c1-600; cut -c1-330 gpurun_out/${T}_bench_b128.log; tail -3 gpurun_out/${T}_lab_pf.log; tail -4 gpurun_out/${T}_lab_ty_bwd.log; tail -4 gpurun_out/${T}_lab_ty_fwd.log
