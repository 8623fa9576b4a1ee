#!/bin/bash
# pmc_extra.sh TAG <bench.py arguments...> -- issue-side counters of one bench workload (branches, instruction fetch, scalar and LDS
# issue, waves in flight), collected like tools/pmc_kernels.sh; summary in gpurun_out/pmcx_TAG/summary.json
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcx_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/bench.json" 2> "$OUT/stats.log" || exit 1
i=0
for pass in "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
            "SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VALU"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/pmc_$i" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.log" || echo "pass $i failed"
    echo "pmc pass $i done"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
