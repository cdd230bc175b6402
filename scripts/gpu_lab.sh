mkdir -p gpurun_out
T=r02r
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_effnetv2.py -m gpu -q -x -k "pw_ or conv or other_activations or model_backward or train_step_equals" 2>&1 | tail -6 | cut -c1-600) > gpurun_out/${T}_tests.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -4 gpurun_out/${T}_tests.log; cut -c1-330 gpurun_out/${T}_bench_b128.log
