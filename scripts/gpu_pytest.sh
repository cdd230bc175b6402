# GPU parity tests only.  usage (via gpurun): bash scripts/gpu_pytest.sh TAG "pytest args"
mkdir -p gpurun_out
T=${1:-pt}; shift
(timeout 1500 python -m pytest -m gpu -q "$@" 2>&1 | tail -150) > gpurun_out/${T}_pytest.log
tail -150 gpurun_out/${T}_pytest.log
