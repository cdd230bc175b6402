# Round-4 lab driver (one gpurun call = one box: batch everything into it).
# usage: gpu_lab.sh TAG ; the sections are chosen by environment variables:
#   LAB_PYTEST="-k expr ..."        pytest -m gpu arguments (empty: skipped); LAB_PYTEST_K="a or b": a -k expression with spaces
#   LAB_STEPS=20                    timed steps of every bench run (default 5)
#   LAB_KERNEL="entry|layers|VAR=a,b;entry|layers|VAR=a,b"   scripts/kernel_lab.py runs
#   LAB_BENCH="NAME:VAR=val VAR2=val;NAME2:"                bench.py runs (5 steps) with a per-launch table each
mkdir -p gpurun_out
T=${1:-lab}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
if [ -n "$LAB_PYTEST$LAB_PYTEST_K" ]; then
  (timeout 1500 python -m pytest tests -m gpu -x -q $LAB_PYTEST ${LAB_PYTEST_K:+-k "$LAB_PYTEST_K"} 2>&1 | cut -c1-3000 | tail -25) > gpurun_out/${T}_pytest.log
  tail -8 gpurun_out/${T}_pytest.log | cut -c1-1200
fi
if [ -n "$LAB_KERNEL" ]; then
  IFS=';' read -ra RUNS <<< "$LAB_KERNEL"
  i=0
  for r in "${RUNS[@]}"; do
    IFS='|' read -r entry layers ab <<< "$r"
    (timeout 600 python scripts/kernel_lab.py --entry $entry --layers $layers ${ab:+--ab $ab} 2>&1 | tail -80) > gpurun_out/${T}_lab${i}.log
    echo "== lab $entry $layers $ab"; cat gpurun_out/${T}_lab${i}.log | cut -c1-150
    i=$((i+1))
  done
fi
if [ -n "$LAB_BENCH" ]; then
  IFS=';' read -ra RUNS <<< "$LAB_BENCH"
  for v in "${RUNS[@]}"; do
    name=${v%%:*}; envs=${v#*:}
    (env $envs timeout ${LAB_TIMEOUT:-600} python bench.py --steps ${LAB_STEPS:-5} --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_${name}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_${name}_bench.log
    echo "$name: $(python -c "
import json
try:
  d=json.loads(open('gpurun_out/${T}_${name}_bench.log').read().strip().splitlines()[-1]); print(round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms', d['roofline']['kernel'], round(d['roofline']['frac'],4))
except Exception as e: print('FAILED', e, open('gpurun_out/${T}_${name}_bench.log').read()[-800:])
")"
  done
fi
