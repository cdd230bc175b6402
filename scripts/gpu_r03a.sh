# r03a: first hardware contact of round 3 -- new parity modules (side configs, V2 bf16 scheme, deterministic SE), bench.
mkdir -p gpurun_out
T=${1:-r03a}
export TMPDIR=/tmp
(free -g | head -2; nproc; rocm-smi --showmeminfo vram 2>/dev/null | tail -3) > gpurun_out/${T}_box.log 2>&1
(timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py -k "squeeze_excite" tests/test_abi.py 2>&1 | tail -8) > gpurun_out/${T}_se.log
(timeout 1500 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py 2>&1 | grep -v "^$" | cut -c1-1500 | tail -150) > gpurun_out/${T}_side.log
(timeout 900 python -m pytest -m gpu -q -s tests/test_effnetv2.py -k "forward or conditioning" 2>&1 | grep -v "^$" | cut -c1-1200 | tail -60) > gpurun_out/${T}_v2.log
(timeout 900 python -m pytest -m gpu -q -s tests/test_gpu_bench_shapes.py -k "stem_se or inference_forward or d0_512" tests/test_gpu_network.py -k "graph_replay or bench_spawns or inference_forward_matches" 2>&1 | grep -v "^$" | cut -c1-1200 | tail -40) > gpurun_out/${T}_misc.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -2) > gpurun_out/${T}_bench_b128.log
cat gpurun_out/${T}_box.log; tail -3 gpurun_out/${T}_se.log; tail -30 gpurun_out/${T}_side.log | cut -c1-600; tail -12 gpurun_out/${T}_v2.log | cut -c1-600; tail -8 gpurun_out/${T}_misc.log | cut -c1-400; cut -c1-700 gpurun_out/${T}_bench_b128.log
