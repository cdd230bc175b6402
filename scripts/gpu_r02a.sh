# round 2, GPU call A: first run of the merged device features and of the new bench-shape parity tests
mkdir -p gpurun_out
T=r02a
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_labeling.py tests/test_preprocess.py tests/test_abi.py -m gpu -q 2>&1 | tail -15) > gpurun_out/${T}_pytest_feat.log
(timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s --durations=8 2>&1 | grep -v "^$" | tail -120) > gpurun_out/${T}_pytest_shapes.log
(timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -s -k "bf16 or two_replicas or hswish or relu6 or graph_replay" 2>&1 | grep -v "^$" | tail -80) > gpurun_out/${T}_pytest_net.log
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tiled or balanced or batchnorm or squeeze" 2>&1 | tail -15) > gpurun_out/${T}_pytest_kern.log
(timeout 600 python bench.py --steps 10 --warmup 3 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -2) > gpurun_out/${T}_bench_b128.log
(timeout 120 python scripts/bench_labeling.py 2>&1 | tail -2) > gpurun_out/${T}_labeling.log
(timeout 200 python scripts/kernel_lab.py --entry pw_bwd_weight --layers mid --ab EDET_WG_BALANCED=0,1 2>&1 | tail -30) > gpurun_out/${T}_lab_wg.log
tail -5 gpurun_out/${T}_pytest_feat.log; tail -8 gpurun_out/${T}_pytest_shapes.log; tail -5 gpurun_out/${T}_pytest_net.log; tail -3 gpurun_out/${T}_pytest_kern.log; cut -c1-400 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_labeling.log; tail -3 gpurun_out/${T}_lab_wg.log
