# Round-5 lab call 2: BatchNorm finalize variants built at compile time (EDET_FIN_SL / EDET_FIN_DEEP in bn_se.hip; variant
# libraries linked beforehand as automl_amd/csrc/build/lab_libedet_hip_v*.so): same-box bench lines + the BatchNorm kernel
# tests under each variant.  The box's tree is a scratch copy: the variant library is copied over the product library there.
mkdir -p gpurun_out
T=${1:-r05lab2}
export TMPDIR=/tmp
cp automl_amd/libedet_hip.so /tmp/libedet_hip_default.so
line() { python -c "
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value'],1),'img/s', round(d['ms_per_step'],3),'ms')
except Exception as e: print('FAILED', e, open(sys.argv[1]).read()[-600:])
" $1; }
for v in default 64_0 64_1 32_1 default; do
  if [ $v = default ]; then cp /tmp/libedet_hip_default.so automl_amd/libedet_hip.so; else cp automl_amd/csrc/build/lab_libedet_hip_v$v.so automl_amd/libedet_hip.so; fi
  (timeout 400 python bench.py --steps 30 --warmup 3 --no_cpu_baseline --no_other_configs 2>&1 | tail -1) > gpurun_out/${T}_bench_fin_$v.log
  echo "finalize variant $v: $(line gpurun_out/${T}_bench_fin_$v.log)"
  if [ $v != default ]; then
    (timeout 300 python -m pytest -m gpu -q -p no:cacheprovider --tb=short tests/test_gpu_kernels.py -k "batchnorm or squeeze or test_dw_" 2>&1 | tail -3 | cut -c1-300) > gpurun_out/${T}_pytest_fin_$v.log
    tail -1 gpurun_out/${T}_pytest_fin_$v.log
  fi
done
cp /tmp/libedet_hip_default.so automl_amd/libedet_hip.so
