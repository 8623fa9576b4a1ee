#!/bin/bash
# pmc_quick.sh TAG <bench.py arguments...> -- the three counter passes that say how many instructions a kernel issued and how its waves' cycles
# divide (a short form of tools/pmc_kernels.sh for A/B questions; summary in gpurun_out/pmcq_TAG/summary.json)
set -o pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcq_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/bench.json" 2> "$OUT/stats.log" || exit 1
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/pmc_$i" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.log" || echo "pass $i failed"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
