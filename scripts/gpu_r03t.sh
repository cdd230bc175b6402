# r03t: stem weight gradient on the matrix cores.
mkdir -p gpurun_out
T=${1:-r03t}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
(timeout 900 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "test_stem" 2>&1 | cut -c1-3000 | tail -25) > gpurun_out/${T}_kern.log
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_network.py tests/test_gpu_bench_shapes.py 2>&1 | cut -c1-3000 | tail -12) > gpurun_out/${T}_net.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -8 gpurun_out/${T}_kern.log | cut -c1-1500; tail -4 gpurun_out/${T}_net.log | cut -c1-1500; grep stem gpurun_out/${T}_launches.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/${T}_bench_b128.log
