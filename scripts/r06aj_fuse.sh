# r06aj: fusion kernels with 32-bit index arithmetic: parity subset + bench lines (D0, D7x)
mkdir -p gpurun_out; T=r06aj; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_gpu_bench_shapes.py -x -q -m gpu -k "fuse or fpn or bit_reproducible or batch128 or oracle_fp32" 2>&1 | grep -E "passed|failed|Error" | tail -3) > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches.txt 2>&1 | tail -1) > gpurun_out/${T}_bench.log
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench.log').read().strip().splitlines()[-1]); print('bench', round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms')"
grep "edet_fuse" gpurun_out/${T}_launches.txt | sort -k4 -n -r | head -8
(timeout 900 python bench.py --model efficientdet-d7x --image_size 1536 --batch 8 --steps 5 --warmup 2 --no_cpu_baseline --no_other_configs --dump_launches gpurun_out/${T}_launches_d7x.txt 2>&1 | grep "^{" | tail -1 | cut -c1-260)
grep "edet_fuse" gpurun_out/${T}_launches_d7x.txt | sort -k4 -n -r | head -6
