mkdir -p gpurun_out
T=r02c
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_labeling.py tests/test_preprocess.py -m gpu -q 2>&1 | tail -5) > gpurun_out/${T}_feat.log
(timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k "not at_d0_640_shapes" 2>&1 | grep -a "teacher\|d0-640\|d0-512\|batch 128\|batch-128\|passed\|failed\|Error\|assert" | cut -c1-1200) > gpurun_out/${T}_shapes.log
(timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -s 2>&1 | grep -a "^forward\|teacher\|cosine\|loss values\|oracle {\|passed\|failed\|worst\|mismatch\|FAILED" | cut -c1-1200) > gpurun_out/${T}_net.log
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "pw_bwd_weight" 2>&1 | tail -4) > gpurun_out/${T}_kern.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
cat gpurun_out/${T}_feat.log; cat gpurun_out/${T}_shapes.log | cut -c1-400; tail -12 gpurun_out/${T}_net.log | cut -c1-300; cat gpurun_out/${T}_kern.log; cut -c1-300 gpurun_out/${T}_bench_b128.log
