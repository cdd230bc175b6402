# SQ / HBM counters of ONE entry point at ONE layer shape (seconds of GPU time instead of a whole-step profile).
# usage (via gpurun): bash scripts/gpu_lab_pmc.sh TAG "<kernel_lab.py arguments>"
#   e.g. bash scripts/gpu_lab_pmc.sh wg20 "--entry pw_bwd_weight --shape 128x20x20x1152x192 --reps 3"
# Separate --pmc passes with --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
mkdir -p gpurun_out
T=${1:-lab}; shift
ARGS="$*"
export TMPDIR=/tmp
run() {
  timeout 300 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python scripts/kernel_lab.py $ARGS > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_$1_agg.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
run mfma "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
tail -4 gpurun_out/${T}_sq1.log | cut -c1-160
for p in sq1 sq2 mfma fetch write; do head -6 gpurun_out/${T}_${p}_agg.txt | cut -c1-220; done
