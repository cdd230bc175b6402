# Round-5 lab call 3: the long side chain of the heads (Engine.long_side_chain, EDET_LONG_SIDE): its equivalence test, the
# tests that run whole training steps through train_lib (eager, captured, data-parallel), then same-box bench lines off / on.
mkdir -p gpurun_out
T=${1:-r05lab3}
export TMPDIR=/tmp
(timeout 900 python -m pytest -m gpu -q -p no:cacheprovider -rf --tb=short --durations=5 tests/test_gpu_network.py tests/test_checkpoint.py \
   -k "long_side_chain or graph_replay or two_steps or moving_normalizer or frozen_variables or bit_reproducible or two_replicas or bench_spawns or rccl_path or round_trip or executed_reference_train_step" 2>&1 | cut -c1-1500 | tail -40) > gpurun_out/${T}_pytest.log
tail -12 gpurun_out/${T}_pytest.log | cut -c1-600
for v in 0 1 0 1; do
  (EDET_LONG_SIDE=$v timeout 400 python bench.py --steps 30 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_long_$v.log
  echo "EDET_LONG_SIDE=$v: $(python -c "
import json
try:
  d=json.loads(open('gpurun_out/${T}_bench_long_$v.log').read().strip().splitlines()[-1]); print(round(d['value'],1),'img/s', round(d['ms_per_step'],3),'ms crc', d['config']['param_crc32'], 'loss', d['config']['loss'])
except Exception as e: print('FAILED', e, open('gpurun_out/${T}_bench_long_$v.log').read()[-800:])
")"
done
