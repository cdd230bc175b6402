# The last GPU call of round 5: `pytest tests/ -x -q -m gpu` in full, one process, file order, on the final HEAD (the
# driver's own command), after a smoke run.  The log goes to profiles/ by hand (scripts/collect_evidence.sh is for full passes).
mkdir -p gpurun_out
T=${1:-r05zz}
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/${T}_smoke.log
(timeout 1300 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | cut -c1-3000 | tail -150) > gpurun_out/${T}_pytest_gpu.log
cat gpurun_out/${T}_smoke.log; tail -16 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
