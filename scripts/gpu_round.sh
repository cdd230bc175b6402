mkdir -p gpurun_out
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -8) > gpurun_out/r1_smoke.log
export TMPDIR=/tmp
(timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline 2>&1 | tail -5) > gpurun_out/r1_bench_b128.log
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b128 -o r1 --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline 2>&1 | tail -5) > gpurun_out/r1_bench_b128_prof.log
find gpurun_out/prof_b128 -name "*kernel_trace.csv" -size +30M -delete
ls -la gpurun_out/prof_b128/* | head
cat gpurun_out/r1_smoke.log gpurun_out/r1_bench_b128.log
