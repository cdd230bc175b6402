# r03h: workgroup caps of the streaming pointwise kernels against the resident-workgroup count (lab), focal parity.
mkdir -p gpurun_out
T=${1:-r03h}
export TMPDIR=/tmp
export EDET_SKIP_SLOW=1
L="timeout 400 python scripts/kernel_lab.py"
(timeout 600 python -m pytest -m gpu -q tests/test_gpu_kernels.py -k "detection_loss" 2>&1 | cut -c1-1500 | tail -6) > gpurun_out/${T}_kern.log
($L --entry pw_fwd --layers all --ab EDET_PWS_FWD_CAP=1024,768,512,256 2>&1 | tail -110) > gpurun_out/${T}_lab_fwdcap.log
($L --entry pw_bwd --layers all --ab EDET_PWS_BWD_CAP=1024,768,512,256 2>&1 | tail -110) > gpurun_out/${T}_lab_bwdcap.log
($L --entry dw_fwd --layers all --ab EDET_DWM_P=unset,1024,1536,2048,3072,4096,8192 2>&1 | tail -130) > gpurun_out/${T}_lab_dwp_fwd.log
($L --entry dw_bwd --layers all --ab EDET_DWM_P=unset,1024,1536,2048,3072,4096,8192 2>&1 | tail -130) > gpurun_out/${T}_lab_dwp_bwd.log
(timeout 600 python -m pytest -m gpu -q -s tests/test_gpu_side_configs.py -k "batch8_train" 2>&1 | grep -v "^$" | cut -c1-1800 | tail -6) > gpurun_out/${T}_side.log
grep TOTAL gpurun_out/${T}_lab_*.log; tail -3 gpurun_out/${T}_kern.log | cut -c1-800; tail -5 gpurun_out/${T}_side.log | cut -c1-1500
