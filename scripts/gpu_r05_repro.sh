# Cross-box reproducibility probe (no code under test changes): the same three commands on any box must print the same
# param_crc32 -- the training step holds no floating-point atomics in either storage type, so the variables after N steps
# are a function of the code and the seeds only.  Also: the data-parallel launch structure at world size 1 on RCCL with the
# all-reduce's own duration.  usage (via gpurun): bash scripts/gpu_r05_repro.sh TAG
mkdir -p gpurun_out
T=${1:-r05repro}
export TMPDIR=/tmp
(timeout 400 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_bf16.log
(timeout 400 python bench.py --dtype f32 --batch 8 --image_size 256 --steps 3 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_f32_256_b8.log
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --force_dist --steps 20 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | grep "^{" | tail -1) > gpurun_out/${T}_bench_force_dist.log
python - gpurun_out/${T}_bench_bf16.log gpurun_out/${T}_bench_f32_256_b8.log gpurun_out/${T}_bench_force_dist.log <<'PY'
import json, sys
for f in sys.argv[1:]:
  try:
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    c = d['config']
    print(f.split('/')[-1], round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms crc32', c['param_crc32'], 'loss', c['loss'],
          'allreduce', c.get('allreduce_ms_per_step'), 'ranks', c.get('per_rank_ms_per_step'))
  except Exception as e:
    print(f, 'FAILED', e, open(f).read()[-500:])
PY
