# round 2, call h: tuned kernels with relu/relu6/hswish, checkpoint round trip on the device, recalibrated tests
mkdir -p gpurun_out
T=r02h
export TMPDIR=/tmp
(timeout 300 python -m pytest "tests/test_gpu_network.py::test_train_step_matches_oracle_fp32" -m gpu -q -x -k "d1" 2>&1 | grep -v "^$" | cut -c1-4000 | tail -40) > gpurun_out/${T}_d1.log
(timeout 1200 python -m pytest tests -m gpu -q --durations=5 2>&1 | cut -c1-2500 | tail -60) > gpurun_out/${T}_pytest_gpu.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -5 gpurun_out/${T}_d1.log | cut -c1-1500; tail -8 gpurun_out/${T}_pytest_gpu.log | cut -c1-400; cut -c1-330 gpurun_out/${T}_bench_b128.log
