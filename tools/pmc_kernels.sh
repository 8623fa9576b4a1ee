#!/bin/bash
# pmc_kernels.sh TAG <bench.py arguments...> -- kernel trace + SQ counter passes of one bench workload, all kernels
# (run ON the GPU box from the repo root; summarise afterwards with tools/pmc_summary.py gpurun_out/pmc_TAG).
# Counters are collected in their own passes (--pmc with --kernel-trace only), the program itself after `--`.
# --ramp-spp 0: the untimed clock ramp of bench.py launches the workload's OWN sample count, so that every dispatch of a kernel in the
# profile is a dispatch of the workload and per-call means describe it (round 4's C5 profiles averaged 72 ramp launches of 64 spp with 5
# of 512: VERDICT r04, weak 6).  pmc_summary.py prints n / min / max per kernel; valu_roofline.py refuses a mean that is not the bench's.
set -o pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --ramp-spp 0 --no-cpu-baseline --no-also > "$OUT/bench.json" 2> "$OUT/stats.log" || exit 1
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/pmc_$i" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --ramp-spp 0 --no-cpu-baseline --no-also > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.log" || echo "pass $i failed"
    echo "pmc pass $i done"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"; cat "$OUT/summary.json" | head -80
