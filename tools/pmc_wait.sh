#!/bin/bash
# pmc_wait.sh TAG <bench.py arguments...> -- where the waves of one bench workload WAIT: in-flight levels of the memory pipes, FIFO-full
# stalls, per-pipe active cycles (collected like tools/pmc_kernels.sh; summary in gpurun_out/pmcw_TAG/summary.json)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcw_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/bench.json" 2> "$OUT/stats.log" || exit 1
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VMEM" \
            "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH_LEVEL" \
            "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/pmc_$i" --output-format csv -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.log" || echo "pass $i failed"
    echo "pmc pass $i done"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
