# r03b: fixed parity tests of the round (side configs incl. the slow D7x teacher-forced step, training preprocessing,
# mish / srelu, SE), rocprof kernel statistics + launch tables of the two side configurations.
mkdir -p gpurun_out
T=${1:-r03b}
export TMPDIR=/tmp
(timeout 1200 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "d7x" 2>&1 | grep -v "^$" | cut -c1-1500 | tail -60) > gpurun_out/${T}_side.log
(timeout 900 python -m pytest -m gpu -q tests/test_preprocess.py tests/test_gpu_kernels.py -k "preprocess or squeeze_excite or other_activations or DetectionInput or training_preprocessing" 2>&1 | cut -c1-1500 | tail -40) > gpurun_out/${T}_kern.log
(timeout 900 python -m pytest -m gpu -q tests/test_gpu_network.py -k "mish or srelu" 2>&1 | cut -c1-1500 | tail -30) > gpurun_out/${T}_act.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_v2s -o v2s --output-format csv -- python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_launches_v2s_224_b256.txt 2>&1 | tail -2) > gpurun_out/${T}_v2s.log
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_d7x -o d7x --output-format csv -- python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x_1536_b8.txt 2>&1 | tail -2) > gpurun_out/${T}_d7x.log
find gpurun_out/prof_${T}_v2s gpurun_out/prof_${T}_d7x -name "*kernel_trace.csv" -delete
tail -25 gpurun_out/${T}_side.log | cut -c1-700; tail -12 gpurun_out/${T}_kern.log | cut -c1-400; tail -8 gpurun_out/${T}_act.log | cut -c1-400; cut -c1-600 gpurun_out/${T}_v2s.log; cut -c1-400 gpurun_out/${T}_d7x.log
