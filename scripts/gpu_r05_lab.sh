# Round-5 lab call: the two tests the serial run r05b had red (fixed), the fusion kernels after the k_fuse refactor, the
# benchmark-shape module (its coverage test reads profiles/CURRENT), then the heads' chain assignment A/B
# (EDET_SIDE_MOVE) as bench lines + a parity subset with the switch on.  usage (via gpurun): bash scripts/gpu_r05_lab.sh TAG
mkdir -p gpurun_out
T=${1:-r05lab}
export TMPDIR=/tmp
(timeout 900 python -m pytest -m gpu -q -p no:cacheprovider -rf --tb=short --durations=5 \
   tests/test_effnetv2.py::test_model_backward_bf16_deferred_reductions_equal_immediate_ones \
   tests/test_gpu_bench_shapes.py "tests/test_gpu_kernels.py::test_fuse" 2>&1 | cut -c1-1200 | tail -40) > gpurun_out/${T}_pytest_a.log
(timeout 600 python -m pytest -m gpu -q -p no:cacheprovider -rf --tb=short tests/test_gpu_kernels.py tests/test_gpu_network.py -k "fuse or channel_fastattn or bit_reproducible" 2>&1 | cut -c1-1200 | tail -25) > gpurun_out/${T}_pytest_b.log
for v in "" class box; do
  (EDET_SIDE_MOVE=$v timeout 400 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_move_${v:-none}.log
  echo "EDET_SIDE_MOVE='$v': $(python -c "
import json
try:
  d=json.loads(open('gpurun_out/${T}_bench_move_${v:-none}.log').read().strip().splitlines()[-1]); print(round(d['value'],1),'img/s', round(d['ms_per_step'],3),'ms  host first/min', round(d['config']['host_enqueue_ms_first_step'],2), round(d['config']['host_enqueue_ms_min'],2))
except Exception as e: print('FAILED', e, open('gpurun_out/${T}_bench_move_${v:-none}.log').read()[-600:])
")"
done
(EDET_SIDE_MOVE=class timeout 600 python -m pytest -m gpu -q -p no:cacheprovider -rf --tb=short tests/test_gpu_network.py tests/test_gpu_bench_shapes.py -k "(train_step_matches_oracle_fp32 and d0 and 128 and not act_type and not fpn and not max_level) or bit_reproducible or d0_640_batch2 or graph_replay" 2>&1 | cut -c1-1200 | tail -15) > gpurun_out/${T}_pytest_move_class.log
tail -12 gpurun_out/${T}_pytest_a.log | cut -c1-500; tail -6 gpurun_out/${T}_pytest_b.log | cut -c1-500; tail -6 gpurun_out/${T}_pytest_move_class.log | cut -c1-500
