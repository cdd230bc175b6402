# One GPU round: parity tests, smoke, bench (+ per-launch table), rocprof kernel stats.  Run via gpurun.
mkdir -p gpurun_out
T=${1:-r1}
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/${T}_pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > gpurun_out/${T}_smoke.log
export TMPDIR=/tmp
(timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
if [ "${2:-}" = "prof" ]; then
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
rm -f gpurun_out/prof_${T}/*kernel_trace.csv
fi
tail -4 gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_smoke.log; cut -c1-600 gpurun_out/${T}_bench_b128.log
