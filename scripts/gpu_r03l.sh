# r03l: dynamic instruction counts of the one-pass pointwise backward at two layer shapes (SQ passes on the lab).
mkdir -p gpurun_out
T=${1:-r03l}
export TMPDIR=/tmp
run() {  # tag counters lab-args
  timeout 200 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/${T}_$1 -o $1 --output-format csv -- python scripts/kernel_lab.py $3 > gpurun_out/${T}_$1.log 2>&1
  python scripts/pmc_agg.py gpurun_out/${T}_$1 > gpurun_out/${T}_$1_agg.txt 2>&1
  rm -rf gpurun_out/${T}_$1
}
for L in b2_project b1_expand b0_project; do
  run ${L}_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "--entry pw_bwd --layers $L --reps 3 --rounds 1"
  run ${L}_sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "--entry pw_bwd --layers $L --reps 3 --rounds 1"
done
for f in gpurun_out/${T}_*_agg.txt; do echo $f; grep -i "fused\|kernel " $f | cut -c1-230; done
