mkdir -p gpurun_out
T=${1:-r02d}
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "test_pw_bwd[" 2>&1 | tail -25) > gpurun_out/${T}_kern.log
(timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -k "one_call" 2>&1 | tail -25) > gpurun_out/${T}_shapes.log
(timeout 400 python scripts/kernel_lab.py --entry pw_bwd --layers all --ab EDET_PW_IMPL=auto,big 2>&1 | tail -60) > gpurun_out/${T}_lab.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -6 gpurun_out/${T}_kern.log; tail -6 gpurun_out/${T}_shapes.log; cat gpurun_out/${T}_lab.log | cut -c1-110; cut -c1-400 gpurun_out/${T}_bench_b128.log
