# bench (+ per-launch table) and rocprof kernel stats only (no pytest).  usage: gpu_bench_prof.sh TAG
mkdir -p gpurun_out
T=${1:-bp}
export TMPDIR=/tmp
(timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench_b128.log
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${T} -o ${T} --output-format csv -- python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs 2>&1 | tail -3) > gpurun_out/${T}_prof.log
rm -f gpurun_out/prof_${T}/*kernel_trace.csv
cut -c1-700 gpurun_out/${T}_bench_b128.log
