# Round-5 lab (NOT to be repeated: four processes on one GPU ran 3x slower than the serial suite, LABNOTES.md): the whole GPU suite over N worker processes (pytest-xdist), every failure
# listed (no -x), then a short bench line.  usage (via gpurun): bash scripts/gpu_r05_suite.sh TAG [workers]
mkdir -p gpurun_out
T=${1:-r05suite}
W=${2:-4}
export TMPDIR=/tmp
nproc > gpurun_out/${T}_nproc.txt
(timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -q -n $W -p no:cacheprovider --durations=12 ${SUITE_ARGS} 2>&1 | cut -c1-2500 | tail -120) > gpurun_out/${T}_pytest.log
tail -60 gpurun_out/${T}_pytest.log | cut -c1-600
if [ -z "$NO_BENCH" ]; then
  (timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench.log
  cut -c1-1500 gpurun_out/${T}_bench.log
fi
