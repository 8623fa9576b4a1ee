#!/bin/bash
# GPU pass: bench lines of the Streams forms
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02b; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
for args in "--scene glass --algorithm streams" "--scene glass --algorithm streams --streams-form stream" "--algorithm streams --streams-form stream" "--algorithm streams"; do
  timeout -k 10 120 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>> "$OUT/bench.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], d['roofline']['kernel'], d['ms_per_step'])" | tee -a "$OUT/bench_streams.txt"
done
