# r06ad: full -m gpu suite on the tree with the network-level ABI (engine clears / joins through the ABI now), a bench line,
# then the EDET_BIG_TPW sweep of r06ac
mkdir -p gpurun_out; T=r06ad; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=5 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | cut -c1-2000 | tail -40) > gpurun_out/${T}_pytest_gpu.log
tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench.log
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench.log').read().strip().splitlines()[-1]); print('bench', round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms')"
bash scripts/r06ac_tpw.sh
