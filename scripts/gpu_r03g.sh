# r03g: focal loss rewrite (parity + lab-less timing through the bench table), weight-gradient split rounding, dispatch
# thresholds; D7x batch-8 test.
mkdir -p gpurun_out
T=${1:-r03g}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 600 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "detection_loss or test_pw_bwd_weight or test_pw_bwd or test_pw_fwd" 2>&1 | cut -c1-1500 | tail -10) > gpurun_out/${T}_kern.log
(timeout 900 python -m pytest -m gpu -q -s tests/test_gpu_bench_shapes.py -k "batch2_train_step or layer_by_layer" 2>&1 | grep -v "^$" | cut -c1-900 | tail -14) > gpurun_out/${T}_net.log
($L --entry pw_bwd_weight --layers all --ab EDET_WGRAD_WGS=unset,480,512,960,1024 2>&1 | tail -130) > gpurun_out/${T}_lab_wgs.log
($L --entry pw_bwd_weight --layers all --ab EDET_PW_IMPL=auto,big,stream 2>&1 | tail -80) > gpurun_out/${T}_lab_impl_wgrad.log
(timeout 600 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "batch8_train" 2>&1 | grep -v "^$" | cut -c1-1800 | tail -8) > gpurun_out/${T}_side.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 600 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x.txt 2>&1 | tail -1 | cut -c1-300) > gpurun_out/${T}_bench_d7x.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -3 gpurun_out/${T}_kern.log | cut -c1-800; tail -8 gpurun_out/${T}_net.log | cut -c1-500; tail -6 gpurun_out/${T}_side.log | cut -c1-1500; cut -c1-330 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_d7x.log; grep "focal\|TOTAL" gpurun_out/${T}_launches.txt
