# r03e: second lab sweep (implementation choice per layer, row tiles per workgroup of the tiled kernels), parity of the
# changed kernels, benches with the r03d defaults.
mkdir -p gpurun_out
T=${1:-r03e}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
($L --entry pw_fwd --layers all --ab EDET_PW_IMPL=auto,big,stream 2>&1 | tail -90) > gpurun_out/${T}_lab_impl_fwd.log
($L --entry pw_bwd_data --layers all --ab EDET_PW_IMPL=auto,big,stream 2>&1 | tail -90) > gpurun_out/${T}_lab_impl_dgrad.log
($L --entry pw_bwd_weight --layers all --ab EDET_PW_IMPL=auto,big,stream 2>&1 | tail -90) > gpurun_out/${T}_lab_impl_wgrad.log
($L --entry pw_fwd --layers mid --ab EDET_BIG_TPW=1,2,4,8 2>&1 | tail -80) > gpurun_out/${T}_lab_tpw_fwd.log
($L --entry pw_bwd_data --layers mid --ab EDET_BIG_TPW=1,2,4,8 2>&1 | tail -80) > gpurun_out/${T}_lab_tpw_dgrad.log
(timeout 900 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "squeeze_excite or test_dw or test_pw_bwd" 2>&1 | cut -c1-1200 | tail -12) > gpurun_out/${T}_kern.log
(timeout 900 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "batch8_train or v2s_224_batch" 2>&1 | grep -v "^$" | cut -c1-1800 | tail -30) > gpurun_out/${T}_side.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 600 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x.txt 2>&1 | tail -1 | cut -c1-300) > gpurun_out/${T}_bench_d7x.log
(timeout 300 python scripts/bench_v2s.py --steps 20 --dump_launches gpurun_out/${T}_launches_v2s.txt 2>&1 | tail -1 | cut -c1-600) > gpurun_out/${T}_bench_v2s.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -4 gpurun_out/${T}_kern.log | cut -c1-800; tail -12 gpurun_out/${T}_side.log | cut -c1-1500; cut -c1-330 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_d7x.log; cat gpurun_out/${T}_bench_v2s.log
