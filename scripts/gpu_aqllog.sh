# ROCclr dispatch log of bench steps: the AQL packets (header, barrier packets, kernarg address, signals) of every kernel.
# usage: [AQL_GRAPH=1 AQL_WARMUP=3] gpu_aqllog.sh TAG -> gpurun_out/TAG_aql.log.gz (the last 9000 relevant lines, uncut)
mkdir -p gpurun_out
T=${1:-aql}
export TMPDIR=/tmp
EDET_GRAPH=${AQL_GRAPH:-0} AMD_LOG_LEVEL=4 python bench.py --steps 1 --warmup ${AQL_WARMUP:-1} --no_cpu_baseline --no_other_configs > /tmp/aql_out.log 2> /tmp/aql_err.log
grep -E "ShaderName|Header|barrier|Barrier" /tmp/aql_err.log | tail -9000 | gzip > gpurun_out/${T}_aql.log.gz
grep -c "" /tmp/aql_err.log
ls -la gpurun_out/${T}_aql.log.gz
