# A/B timing of bench.py under different environment settings, after selected parity tests.
# usage: gpu_ab.sh TAG "pytest -k expr" "NAME1:VAR=val VAR2=val" "NAME2:..." ...   (NAME: with no vars = default)
mkdir -p gpurun_out
T=${1:-ab}; K=${2:-pw}; shift 2
if [ -n "$K" ]; then
  (timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -15) > gpurun_out/${T}_pytest.log
  tail -4 gpurun_out/${T}_pytest.log
fi
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  (env $envs timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_${name}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_${name}_bench.log
  echo "$name: $(python -c "
import json,sys
try:
  d=json.loads(open('gpurun_out/${T}_${name}_bench.log').read().strip().splitlines()[-1]); print(round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms', d['roofline']['kernel'], round(d['roofline']['frac'],4))
except Exception as e: print('FAILED', e, open('gpurun_out/${T}_${name}_bench.log').read()[-600:])
")"
done
