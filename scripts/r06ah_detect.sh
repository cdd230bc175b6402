# r06ah: the detect program (edet_detect) and the extended plan tests on the device
mkdir -p gpurun_out; T=r06ah; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_gpu_bench_shapes.py -x -q -m gpu -k "plan or detect or replay or c_host or plumbing" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 | cut -c1-500) > gpurun_out/${T}_pytest.log; tail -30 gpurun_out/${T}_pytest.log
