# Full GPU evidence pass: parity tests, smoke, bench (with cpu_baseline), rocprof kernel stats, HBM PMC passes.
# usage (via gpurun): [SKIP_PMC=1] bash scripts/gpu_full.sh TAG
mkdir -p gpurun_out
T=${1:-full}
STEPS_IN_PMC_RUN=6
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -25) > gpurun_out/${T}_pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > gpurun_out/${T}_smoke.log
(timeout 600 python bench.py --steps 10 --warmup 3 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
rm -f gpurun_out/prof_${T}/*kernel_trace.csv gpurun_out/prof_${T}/*/*kernel_trace.csv
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
if [ -z "$SKIP_PMC" ]; then
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
fi
tail -6 gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_smoke.log; cut -c1-900 gpurun_out/${T}_bench_b128.log; head -5 gpurun_out/${T}_pmc_fetch.txt
