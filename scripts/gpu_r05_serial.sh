# Round-5: the whole GPU suite the way the driver runs it (one process, file order), but WITHOUT -x: every failure is
# listed with a short traceback.  usage (via gpurun): bash scripts/gpu_r05_serial.sh TAG ["extra pytest args"]
mkdir -p gpurun_out
T=${1:-r05serial}
export TMPDIR=/tmp
(timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -q -p no:cacheprovider -rf --tb=short --durations=12 $2 2>&1 | cut -c1-1500) > gpurun_out/${T}_pytest_full.log
grep -v "^\.*s*[.s]* *\[" gpurun_out/${T}_pytest_full.log | tail -150 | cut -c1-700
tail -3 gpurun_out/${T}_pytest_full.log
