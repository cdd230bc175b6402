# quick GPU check: selected parity tests + bench with per-launch table.  usage: gpu_quick.sh TAG "pytest -k expr"
mkdir -p gpurun_out
T=${1:-q}
K=${2:-pw_fwd}
(timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -15) > gpurun_out/${T}_pytest.log
(timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -3) > gpurun_out/${T}_bench.log
tail -6 gpurun_out/${T}_pytest.log; cut -c1-330 gpurun_out/${T}_bench.log
