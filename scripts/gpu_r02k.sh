# round 2, call k: full -m gpu run + bench on the tree with the r02i/j lab winners hardwired and the SE / BN streaming loops unrolled
mkdir -p gpurun_out
T=r02k
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --durations=5 2>&1 | cut -c1-3000 | tail -150) > gpurun_out/${T}_pytest_gpu.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
tail -8 gpurun_out/${T}_pytest_gpu.log | cut -c1-400; cut -c1-330 gpurun_out/${T}_bench_b128.log
