#!/bin/bash
# profile_round.sh -- the measurement pass behind profiles/<tag>_* (run ON the GPU box, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
# then, back in the container:  python tools/summarize_pmc.py gpurun_out/prof_r01 r01  and copy the rest (see below).
# Counters are collected in their own passes (rocprofv3 --pmc with --kernel-trace only), the program itself after `--`.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"

# 1. the bench line as the driver runs it (N = 1, defaults), cpu_baseline included
python3 bench.py > "$OUT/bench.json"
echo "bench done"

# 2. kernel trace of the same command (average duration of the dominant kernel must agree with roofline.kernel_ms)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "kernel trace done"

# 3. PMC passes
i=0
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: "Request exceeds the capabilities of the hardware to collect")
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/pmc_$i" --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.log"
    echo "pmc pass $i done"
done

# 4. side measurements quoted in DESIGN.md
python3 tools/measure_extra.py > "$OUT/extra.json"
python3 tools/measure_host_copies.py > "$OUT/host_copies.log"
echo "all done: $OUT"
