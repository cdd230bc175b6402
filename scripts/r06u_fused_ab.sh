#!/bin/bash
# same-box A/B of the fused MBConv head in the headline step, plus the network tests that exercise it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_network.py -x -q -m gpu -k "bf16 or bit_reproducible or reference_graph" > gpurun_out/r06u_pytest.log 2>&1
tail -5 gpurun_out/r06u_pytest.log
for f in 0 1; do
  EDET_MBCONV_FUSED=$f python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/r06u_launches_f$f.txt > gpurun_out/r06u_bench_f$f.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06u_bench_f$f.log
  grep -E "mbconv|320x320x16->96|160x160x24->144|320x320x96 k3s2|160x160x144 k" gpurun_out/r06u_launches_f$f.txt
done
