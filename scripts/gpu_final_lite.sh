# Evidence pass on the FINAL tree when the GPU budget does not hold the full one (scripts/gpu_final.sh, ~15 min): the same
# rocprof statistics / PMC passes / bench lines, the kernel timeline of a replayed step, and the parity tests of the files
# that the changes since the last full pass touch (LITE_TESTS / LITE_K).  usage (via gpurun): bash scripts/gpu_final_lite.sh TAG
mkdir -p gpurun_out
T=${1:-final}
STEPS_IN_PMC_RUN=6
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > gpurun_out/${T}_smoke.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
find gpurun_out/prof_${T} -name "*kernel_trace.csv" -delete
cp gpurun_out/prof_${T}/${T}_kernel_stats.csv profiles/${T}_kernel_stats_b128.csv && echo ${T}_kernel_stats_b128.csv > profiles/CURRENT
(timeout 900 python -m pytest ${LITE_TESTS:-tests/test_gpu_bench_shapes.py tests/test_gpu_network.py} -m gpu -q --durations=5 2>&1 | cut -c1-3000 | tail -40) > gpurun_out/${T}_pytest_gpu.log
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "${LITE_K:-fuse or batchnorm or test_dw_ or squeeze}" 2>&1 | cut -c1-3000 | tail -12) >> gpurun_out/${T}_pytest_gpu.log
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python scripts/pmc_traffic.py gpurun_out/${T}_pmc_fetch.txt gpurun_out/${T}_pmc_write.txt $STEPS_IN_PMC_RUN profiles/${T}_traffic.json && cp profiles/${T}_traffic.json gpurun_out/
(timeout 900 python bench.py --steps 20 --warmup 5 --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
bash scripts/gpu_trace.sh ${T} > gpurun_out/${T}_trace.log 2>&1
python scripts/timeline_gaps.py gpurun_out/${T}_trace.csv.gz 6 > gpurun_out/${T}_timeline.txt 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_v2s -o v2s --output-format csv -- python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_launches_v2s_224_b256.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_v2s.json
(timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T}_d7x -o d7x --output-format csv -- python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x_1536_b8.txt 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_d7x.json
find gpurun_out/prof_${T}_v2s gpurun_out/prof_${T}_d7x -name "*kernel_trace.csv" -delete
tail -14 gpurun_out/${T}_pytest_gpu.log | cut -c1-300; cat gpurun_out/${T}_smoke.log; cut -c1-1200 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_timeline.txt | head -8; cut -c1-250 gpurun_out/${T}_bench_v2s.json; cut -c1-250 gpurun_out/${T}_bench_d7x.json
