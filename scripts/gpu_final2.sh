# Second leg of the evidence pass (after a change that renames kernel symbols but not code): kernel statistics of the three
# configurations on the final tree, the tests that failed or depend on those statistics, the bench line.
mkdir -p gpurun_out
T=${1:-final2}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
find gpurun_out/prof_${T} -name "*kernel_trace.csv" -delete
cp gpurun_out/prof_${T}/${T}_kernel_stats.csv profiles/${T}_kernel_stats_b128.csv && echo ${T}_kernel_stats_b128.csv > profiles/CURRENT
(timeout 1200 python -m pytest -m gpu -q tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py tests/test_gpu_side_configs.py -k "other_activations or v2s or bench_shapes or test_pw_fwd" 2>&1 | cut -c1-3000 | tail -40) > gpurun_out/${T}_pytest_gpu.log
(timeout 900 python bench.py --steps 20 --warmup 5 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_v2s -o v2s --output-format csv -- python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_launches_v2s_224_b256.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_v2s.json
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_d7x -o d7x --output-format csv -- python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x_1536_b8.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_d7x.json
find gpurun_out/prof_${T}_v2s gpurun_out/prof_${T}_d7x -name "*kernel_trace.csv" -delete
tail -6 gpurun_out/${T}_pytest_gpu.log | cut -c1-600; cut -c1-1500 gpurun_out/${T}_bench_b128.log; cut -c1-300 gpurun_out/${T}_bench_v2s.json; cut -c1-300 gpurun_out/${T}_bench_d7x.json
