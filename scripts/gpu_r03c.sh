# r03c: first hardware contact of the NOY pointwise backward, the sliced SE layers and the ragged-column tiled kernels:
# kernel parity, network parity (teacher-forced d0 step), lab A/B, benches.
mkdir -p gpurun_out
T=${1:-r03c}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
(timeout 900 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "test_pw_bwd or squeeze_excite" 2>&1 | cut -c1-1500 | tail -40) > gpurun_out/${T}_kern.log
(timeout 1200 python -m pytest -m gpu -q -s tests/test_gpu_bench_shapes.py -k "pw_bwd_one_call or layer_by_layer or batch2_train_step or batch128_train" 2>&1 | grep -v "^$" | cut -c1-1200 | tail -40) > gpurun_out/${T}_net.log
(timeout 1200 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "d7x_1536_shapes or batch8" tests/test_gpu_network.py -k "d7x" 2>&1 | grep -v "^$" | cut -c1-1200 | tail -40) > gpurun_out/${T}_side.log
(timeout 300 python scripts/kernel_lab.py --entry pw_bwd --layers big --ab EDET_PW_NOY=1,0 2>&1 | tail -14) > gpurun_out/${T}_lab_noy.log
(timeout 300 python scripts/kernel_lab.py --entry pw_bwd_weight --layers mid --ab EDET_WGRAD_WGS=2048,1024,512,256 2>&1 | tail -80) > gpurun_out/${T}_lab_wgs.log
(EDET_PW_NOY=0 timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1 | cut -c1-400) > gpurun_out/${T}_bench_noy0.log
(timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench_b128.log
(timeout 600 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x.txt 2>&1 | tail -1 | cut -c1-500) > gpurun_out/${T}_bench_d7x.log
tail -6 gpurun_out/${T}_kern.log | cut -c1-600; tail -14 gpurun_out/${T}_net.log | cut -c1-700; tail -8 gpurun_out/${T}_side.log | cut -c1-700; cat gpurun_out/${T}_lab_noy.log; tail -5 gpurun_out/${T}_lab_wgs.log; cat gpurun_out/${T}_bench_noy0.log; cut -c1-400 gpurun_out/${T}_bench_b128.log; cat gpurun_out/${T}_bench_d7x.log
