#!/bin/bash
# full -m gpu suite + the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${1:-r06x}
python -m pytest tests/ -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1
tail -4 gpurun_out/${T}_pytest_gpu.log
( time python bench.py ) > gpurun_out/${T}_bench_default.log 2>&1
tail -c 3000 gpurun_out/${T}_bench_default.log
