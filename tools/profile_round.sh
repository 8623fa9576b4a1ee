#!/bin/bash
# profile_round.sh -- the measurement pass behind profiles/<tag>_* (run ON the GPU box, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_round.sh r05'
# then, back in the container:  bash tools/collect_profiles.sh r05   (copies the summaries into profiles/).
# Counters are collected in their own passes (rocprofv3 --pmc with --kernel-trace only), the program itself after `--`.
set -o pipefail
TAG=${1:-r06}
PART=${2:-all}    # all | a (bench lines, kernel traces, PMC passes: ~10 min) | b (rates, part bounds, statistics builds, traffic terms: ~10 min) -- two gpurun calls of <= 1200 s
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
if [ "$PART" != "b" ]; then rm -rf "$OUT"; fi
mkdir -p "$OUT"   # (locally, delete gpurun_out/pmc_* and gpurun_out/prof_<tag> of earlier runs first: gpurun MERGES what comes back into what is there)
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"

if [ "$PART" != "b" ]; then
# 0. what this profile is a profile OF: the hash of kernel sources, headers and flags (bench.py says `stale` when they have moved on)
python3 -c "import __graft_entry__ as g; print(g.load_package()._build.code_id())" > "$OUT/build_id.txt"
# 1. the bench line as the driver runs it (N = 1, defaults), cpu_baseline included; C4 on one GPU (the N > 1 workload)
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
python3 bench.py --scaling strong --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4_one_gpu.json" 2>> "$OUT/bench.log"; echo "bench C4 rc=$?"
python3 bench.py --scene glass --algorithm streams --streams-form pixel --width 3840 --height 2160 --spp 512 --part-of 8 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c5_part.json" 2>> "$OUT/bench.log"; echo "bench C5 part rc=$?"
python3 bench.py --scene glass --algorithm streams --streams-form stream --width 3840 --height 2160 --spp 512 --part-of 8 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_c5_part_stream.json" 2>> "$OUT/bench.log"; echo "bench C5 part, stream form rc=$?"
# the kernel trace of the driver's own command (default steps): its mean for render_inline_kernel must agree with roofline.kernel_ms
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_default" --output-format csv -- python3 bench.py --no-cpu-baseline --no-also > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats_default.log"; echo "stats rc=$?"

# 2. kernel trace + PMC passes, per workload: the headline kernel and the Streams kernels
#    (tools/pmc_kernels.sh writes gpurun_out/pmc_<name>/; the same bench command under every pass)
bash tools/pmc_kernels.sh c2 > "$OUT/pmc_c2.log" 2>&1; echo "pmc c2 rc=$?"
bash tools/pmc_kernels.sh streams --algorithm streams > "$OUT/pmc_streams.log" 2>&1; echo "pmc streams rc=$?"
bash tools/pmc_kernels.sh glass_tree --scene glass --algorithm streams > "$OUT/pmc_glass_tree.log" 2>&1; echo "pmc glass tree rc=$?"
bash tools/pmc_kernels.sh glass_stream --scene glass --algorithm streams --streams-form stream > "$OUT/pmc_glass_stream.log" 2>&1; echo "pmc glass stream rc=$?"
bash tools/pmc_kernels.sh s16_stream --algorithm streams --streams-form stream > "$OUT/pmc_s16_stream.log" 2>&1; echo "pmc s16 stream rc=$?"
bash tools/pmc_kernels.sh c5_tree --scene glass --algorithm streams --streams-form pixel --width 3840 --height 2160 --spp 512 --part-of 8 > "$OUT/pmc_c5_tree.log" 2>&1; echo "pmc c5 tree rc=$?"
bash tools/pmc_kernels.sh c5_stream --scene glass --algorithm streams --streams-form stream --width 3840 --height 2160 --spp 512 --part-of 8 > "$OUT/pmc_c5_stream.log" 2>&1; echo "pmc c5 stream rc=$?"

fi
if [ "$PART" != "a" ]; then
# 3. per-opcode VALU issue costs (shader-clock domain) -- with 2. the inputs of tools/valu_roofline.py
if [ -x build/valu_rates ]; then timeout -k 10 600 build/valu_rates 1 2 6 8 > "$OUT/valu_rates.json" 2> "$OUT/valu_rates.err"; echo "valu_rates rc=$?"; fi

# 4. strong scaling bound on one GPU (C4 parts), side measurements, round occupancies
timeout -k 10 400 python3 tools/part_bound.py --config c4 > "$OUT/c4_part.json" 2> "$OUT/c4_part.log"; echo "c4_part rc=$?"
timeout -k 10 600 python3 tools/part_bound.py --config c5 > "$OUT/c5_part.json" 2> "$OUT/c5_part.log"; echo "c5_part rc=$?"
timeout -k 10 400 python3 tools/measure_extra.py > "$OUT/extra.json" 2> "$OUT/extra.log"; echo "extra rc=$?"
timeout -k 10 300 python3 tools/phase_stats.py > "$OUT/phase_stats.json" 2> "$OUT/phase_stats.log"; echo "phase rc=$?"
# 5. the stream form: lane participation per block of the split kernel, the end of the pixels kernel's launch; the contracted-arithmetic report
timeout -k 10 300 python3 tools/split_stats.py > "$OUT/split_stats.json" 2> "$OUT/split_stats.log"; echo "split stats rc=$?"
timeout -k 10 300 python3 tools/tree_stats.py > "$OUT/tree_stats.json" 2> "$OUT/tree_stats.log"; echo "tree stats rc=$?"
timeout -k 10 300 python3 tools/ab.py run --workloads glass_tree,glass_stream,glass_stream_uniform,glass_stream_g4,glass_stream_g8,glass_stream_g16,glass_stream_b8,glass_stream_b32,c5_tree,c5_stream,c5_stream_uniform,c5_stream_g8,streams,s16_stream default > "$OUT/ab_options.txt" 2>&1; echo "ab options rc=$?"
timeout -k 10 300 python3 tools/tail_stats.py > "$OUT/tail_stats.json" 2> "$OUT/tail_stats.log"; echo "tail stats rc=$?"
timeout -k 10 300 python3 tools/tail_stats.py phases > "$OUT/tail_phases.json" 2>> "$OUT/tail_stats.log"; echo "tail phases rc=$?"
# 6. issue-side counters of the two forms of render Streams on S16 (branches, instruction fetch, scalar and LDS issue)
bash tools/pmc_extra.sh streams --algorithm streams > "$OUT/pmcx_streams.log" 2>&1; echo "pmcx streams rc=$?"
bash tools/pmc_extra.sh s16_stream --algorithm streams --streams-form stream > "$OUT/pmcx_s16_stream.log" 2>&1; echo "pmcx s16 stream rc=$?"
timeout -k 10 300 python3 tools/contracted_report.py > "$OUT/contracted.json" 2> "$OUT/contracted.log"; echo "contracted rc=$?"
# 7. where the split kernel's HBM bytes go, term by term (product, ticket orders, measurement builds that leave one source out)
# 8a. the ordered passes of the stream form under load, default (fenced) hand-off: every launch bit for bit against the chain kernel
timeout -k 10 400 python3 tools/soak_passes.py 120 fenced > "$OUT/soak_passes.json" 2> "$OUT/soak_passes.log"; echo "soak rc=$?"
# 8. the reference's own configuration through the three forms of the boundary (resident, compatible closure, chained closure)
timeout -k 10 300 python3 tools/c0_calls.py > "$OUT/c0_calls.json" 2> "$OUT/c0_calls.log"; echo "c0 calls rc=$?"
timeout -k 10 900 python3 tools/traffic_terms.py --out "$ROOT/gpurun_out/traffic_terms_$TAG" --variants product,pass_by_pass,one_group,uniform_4x16,skip_item_atomics,skip_ring_atomics,skip_item_costs > "$OUT/traffic_terms.json" 2> "$OUT/traffic_terms.log"; echo "traffic terms rc=$?"
fi
echo "all done ($PART): $OUT"
