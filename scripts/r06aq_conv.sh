# r06aq: 3 x 3 convolution from an LDS-resident halo tile (conv_halo.hip): parity, V2-S forward A/B with per-launch tables
T=${1:-r06aq}
mkdir -p gpurun_out
export TMPDIR=/tmp EDET_SKIP_SLOW=1
(timeout 900 python -m pytest tests/test_effnetv2.py -m gpu -x -q -p no:cacheprovider -k "conv_fwd" 2>&1 | tail -15) > gpurun_out/${T}_pytest.log; tail -6 gpurun_out/${T}_pytest.log | cut -c1-600
for h in 0 1 0 1; do
  echo "== v2s EDET_CONV_HALO=$h"; (EDET_CONV_HALO=$h timeout 300 python scripts/bench_v2s.py --steps 10 --dump_launches gpurun_out/${T}_v2s_h${h}_launches.txt 2>&1 | grep "^{" | tail -1 | cut -c1-200)
done
grep conv_fwd gpurun_out/${T}_v2s_h0_launches.txt; echo; grep conv_fwd gpurun_out/${T}_v2s_h1_launches.txt
