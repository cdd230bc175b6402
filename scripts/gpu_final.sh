# Full GPU evidence pass of a round on the FINAL tree (r05: the measurements first, then `pytest tests/ -x -q -m gpu` in full,
# then the side configurations): parity tests, smoke, rocprof kernel statistics (headline + the two
# side configurations), HBM PMC passes -> traffic json, SQ passes, the bench line (reads the traffic json of this run),
# labelling / post-processing benches.  usage (via gpurun): bash scripts/gpu_final.sh TAG
mkdir -p gpurun_out
T=${1:-final}
STEPS_IN_PMC_RUN=6       # bench.py --steps 1 --warmup 1 in graph mode: 3 warm-up incl. capture, 1 profiled eager, 1 replay, 1 eager roofline step
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > gpurun_out/${T}_smoke.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
find gpurun_out/prof_${T} -name "*kernel_trace.csv" -delete
# the coverage test (tests/test_gpu_bench_shapes.py) reads the kernel statistics that profiles/CURRENT names: they must be
# the ones of THIS tree, so the statistics are taken first and the parity run second
cp gpurun_out/prof_${T}/${T}_kernel_stats.csv profiles/${T}_kernel_stats_b128.csv && echo ${T}_kernel_stats_b128.csv > profiles/CURRENT
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python scripts/pmc_traffic.py gpurun_out/${T}_pmc_fetch.txt gpurun_out/${T}_pmc_write.txt $STEPS_IN_PMC_RUN profiles/${T}_traffic.json && cp profiles/${T}_traffic.json gpurun_out/
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run mfma "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"
(timeout 900 python bench.py --steps 20 --warmup 5 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
bash scripts/gpu_trace.sh ${T} > gpurun_out/${T}_trace.log 2>&1      # kernel timeline of the run -> profiles/${T}_timeline_b128.csv.gz
python scripts/timeline_gaps.py gpurun_out/${T}_trace.csv.gz 6 > gpurun_out/${T}_timeline.txt 2>&1
# the complete parity suite, exactly as the driver runs it (one process, file order, -x), on the tree the evidence above was
# taken from; PYTEST_X= (empty) lists every failure instead of stopping at the first
(timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | cut -c1-3000 | tail -150) > gpurun_out/${T}_pytest_gpu.log
tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_v2s -o v2s --output-format csv -- python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_launches_v2s_224_b256.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_v2s.json
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_d7x -o d7x --output-format csv -- python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x_1536_b8.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_d7x.json
find gpurun_out/prof_${T}_v2s gpurun_out/prof_${T}_d7x -name "*kernel_trace.csv" -delete
(timeout 600 python scripts/bench_plan.py --steps 20 --detect 2>&1 | grep "^{" | tail -2) > gpurun_out/${T}_plan_bench.json      # the step through edet_train_step (C host runtime)
(timeout 120 python scripts/bench_labeling.py 2>&1 | tail -1) > gpurun_out/${T}_labeling.json
(timeout 200 python scripts/bench_postprocess.py 2>&1 | tail -12) > gpurun_out/${T}_postprocess_bench_b128.jsonl
tail -8 gpurun_out/${T}_pytest_gpu.log | cut -c1-400; cat gpurun_out/${T}_smoke.log; cut -c1-1500 gpurun_out/${T}_bench_b128.log; head -4 gpurun_out/${T}_pmc_fetch.txt | cut -c1-200; cut -c1-300 gpurun_out/${T}_bench_v2s.json; cut -c1-300 gpurun_out/${T}_bench_d7x.json; cut -c1-400 gpurun_out/${T}_plan_bench.json; cat gpurun_out/${T}_labeling.json
