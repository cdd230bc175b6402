# Lab r06ab: first contact of the step plan / network-level C ABI with the device
mkdir -p gpurun_out; T=r06ab; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_plan.py -x -q -m "gpu or not gpu" 2>&1 | tail -40) > gpurun_out/${T}_pytest.log; tail -30 gpurun_out/${T}_pytest.log | cut -c1-400
(timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "bit_reproducible or graph_replay or two_steps" 2>&1 | tail -5) > gpurun_out/${T}_pytest2.log; tail -3 gpurun_out/${T}_pytest2.log | cut -c1-300
