# PMC passes over one bench step (separate passes, --kernel-trace only, as gpurun requires)
mkdir -p gpurun_out
T=${1:-pmc}
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|gpu-agent|.*SQ_WAIT|.*SQ_ACTIVE_INST|.*SQ_INSTS_V|.*SQ_BUSY_CY|.*SQ_WAVE_CY|.*FETCH_SIZE|.*WRITE_SIZE|.*TA_BUSY|.*TCP_|.*GRBM_GUI)" | head -80 > gpurun_out/${T}_counters.txt
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_$1_agg.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
run fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
run write "WRITE_SIZE"
tail -3 gpurun_out/${T}_sq1.log | cut -c1-200
head -40 gpurun_out/${T}_sq1_agg.txt
