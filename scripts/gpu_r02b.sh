mkdir -p gpurun_out
T=r02b
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -s -k "forward_matches_oracle_bf16 or tracks_both or (train_step_matches_oracle_fp32 and relu6)" 2>&1 | grep -a "^forward\|vs emulating\|cosine\|loss values\|oracle {\|passed\|failed\|worst\|mismatch" | cut -c1-900) > gpurun_out/${T}_net.log
(timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k "batch2_train_step" 2>&1 | grep -a "d0-640\|passed\|failed\|Error" | cut -c1-900) > gpurun_out/${T}_shapes.log
(timeout 300 python scripts/diag_labeling.py 2>&1 | tail -40) > gpurun_out/${T}_lab.log
(timeout 600 python scripts/diag_tiling.py bf16 128 2>&1 | tail -150) > gpurun_out/${T}_tiling_bf16.log
(timeout 600 python scripts/diag_tiling.py f32 16 2>&1 | tail -60) > gpurun_out/${T}_tiling_f32.log
cat gpurun_out/${T}_net.log | cut -c1-300; cat gpurun_out/${T}_shapes.log | cut -c1-300
