# HBM / SQ counters of the fused MBConv head kernels at one layer (scripts/mbconv_lab.py); separate --pmc passes
mkdir -p gpurun_out
T=${1:-mbf}; shift
ARGS="$*"
export TMPDIR=/tmp
run() {
  timeout 300 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python scripts/mbconv_lab.py $ARGS > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_$1_agg.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
for p in fetch write sq1 sq2; do head -12 gpurun_out/${T}_${p}_agg.txt | cut -c1-230; done
