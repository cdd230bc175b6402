#!/bin/bash
# full -m gpu suite on the current tree, then a same-box A/B of the fused MBConv head in the headline step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${1:-r06w}
python -m pytest tests/ -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1
tail -5 gpurun_out/${T}_pytest_gpu.log
for f in 0 1; do
  EDET_MBCONV_FUSED=$f python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_f$f.txt > gpurun_out/${T}_bench_f$f.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/${T}_bench_f$f.log
done
grep -E "mbconv" gpurun_out/${T}_launches_f1.txt
