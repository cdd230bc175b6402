# SQ counter passes over one bench step (separate passes, --kernel-trace only, as gpurun requires); the HBM passes
# (FETCH_SIZE / WRITE_SIZE) are part of scripts/gpu_full.sh.   usage (via gpurun): bash scripts/gpu_pmc.sh TAG
mkdir -p gpurun_out
T=${1:-pmc}
export TMPDIR=/tmp
run() {
  timeout 600 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
run mfma "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"
tail -3 gpurun_out/${T}_sq1.log | cut -c1-200
head -12 gpurun_out/${T}_pmc_sq1.txt | cut -c1-230; head -8 gpurun_out/${T}_pmc_mfma.txt | cut -c1-200
