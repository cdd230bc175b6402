# r03w: fused depthwise backward with the old gradient (beta) prefetched through the row ring.
mkdir -p gpurun_out
T=${1:-r03w}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "dw_bwd" 2>&1 | cut -c1-2500 | tail -6) > gpurun_out/${T}_kern.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -3 gpurun_out/${T}_kern.log | cut -c1-600; grep "edet_dw_bwd " gpurun_out/${T}_launches.txt | head -12; grep -o '"ms_per_step": [0-9.]*' gpurun_out/${T}_bench_b128.log
