#!/bin/bash
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02c; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/s16_stream" --output-format csv -- python3 bench.py --algorithm streams --streams-form stream --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/a.json" 2> "$OUT/a.log"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/glass_stream" --output-format csv -- python3 bench.py --scene glass --algorithm streams --streams-form stream --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.log"
find "$OUT" -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 "$f"; done
