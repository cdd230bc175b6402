# Full GPU evidence pass of a round: parity tests, smoke, rocprof kernel stats, HBM PMC passes -> traffic json, then the
# bench line (which reads that traffic json), labelling / post-processing benches.
# usage (via gpurun): [SKIP_PMC=1] bash scripts/gpu_full.sh TAG
mkdir -p gpurun_out
T=${1:-full}
STEPS_IN_PMC_RUN=6       # bench.py --steps 1 --warmup 1 in graph mode: 3 warm-up incl. capture, 1 profiled eager, 1 replay, 1 eager roofline step
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --durations=5 2>&1 | cut -c1-3000 | tail -120) > gpurun_out/${T}_pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > gpurun_out/${T}_smoke.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
find gpurun_out/prof_${T} -name "*kernel_trace.csv" -delete
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
if [ -z "$SKIP_PMC" ]; then
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python scripts/pmc_traffic.py gpurun_out/${T}_pmc_fetch.txt gpurun_out/${T}_pmc_write.txt $STEPS_IN_PMC_RUN profiles/${T}_traffic.json && cp profiles/${T}_traffic.json gpurun_out/
fi
(timeout 900 python bench.py --steps 20 --warmup 5 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
(timeout 120 python scripts/bench_labeling.py 2>&1 | tail -1) > gpurun_out/${T}_labeling.json
(timeout 200 python scripts/bench_postprocess.py 2>&1 | tail -12) > gpurun_out/${T}_postprocess_bench_b128.jsonl
tail -6 gpurun_out/${T}_pytest_gpu.log | cut -c1-300; cat gpurun_out/${T}_smoke.log; cut -c1-1200 gpurun_out/${T}_bench_b128.log; head -5 gpurun_out/${T}_pmc_fetch.txt; cat gpurun_out/${T}_labeling.json
