# After `gpurun -- bash scripts/gpu_final.sh TAG` (or gpu_final_lite.sh): copies what the judge should read from the scratch
# directory gpurun_out/ into profiles/ under the round's naming scheme and points profiles/CURRENT at the tag.
# usage: bash scripts/collect_evidence.sh TAG
T=${1:?tag}
G=gpurun_out
P=profiles
cpn() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
python - "$G/${T}_bench_b128.log" "$P/${T}_bench_b128.json" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d = json.loads(lines[-1])
open(sys.argv[2], 'w').write(json.dumps(d, indent=1) + '\n')
print('  %s: %.1f images/s, %.2f ms/step' % (sys.argv[2], d['value'], d['ms_per_step']))
PY
cpn $G/prof_${T}/${T}_kernel_stats.csv $P/${T}_kernel_stats_b128.csv
cpn $G/prof_${T}_v2s/v2s_kernel_stats.csv $P/${T}_kernel_stats_v2s_224_b256.csv
cpn $G/prof_${T}_d7x/d7x_kernel_stats.csv $P/${T}_kernel_stats_d7x_1536_b8.csv
cpn $G/${T}_launches.txt $P/${T}_launch_table_b128.txt
cpn $G/${T}_launches_v2s_224_b256.txt $P/${T}_launch_table_v2s_224_b256.txt
cpn $G/${T}_launches_d7x_1536_b8.txt $P/${T}_launch_table_d7x_1536_b8.txt
for c in fetch write sq1 mfma; do cpn $G/${T}_pmc_$c.txt $P/${T}_pmc_${c}_b128.txt; done
cpn $G/${T}_traffic.json $P/${T}_traffic.json
cpn $G/${T}_trace.csv.gz $P/${T}_timeline_b128.csv.gz
cpn $G/${T}_timeline.txt $P/${T}_timeline_b128.txt
cpn $G/${T}_pytest_gpu.log $P/${T}_pytest_gpu.log
cpn $G/${T}_bench_v2s.json $P/${T}_bench_v2s_224_b256.json
cpn $G/${T}_bench_d7x.json $P/${T}_bench_d7x_1536_b8.json
cpn $G/${T}_plan_bench.json $P/${T}_plan_bench.json
cpn $G/${T}_labeling.json $P/${T}_labeling.json
cpn $G/${T}_postprocess_bench_b128.jsonl $P/${T}_postprocess_bench_b128.jsonl
[ -s $P/${T}_kernel_stats_b128.csv ] && echo ${T}_kernel_stats_b128.csv > $P/CURRENT && echo "  profiles/CURRENT -> $(cat $P/CURRENT)"
