#!/bin/bash
# collect_profiles.sh TAG -- after tools/profile_round.sh ran on the GPU box: summarise and copy into profiles/ (tracked).
set -e
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/prof_$TAG
cp "$SRC/bench.json" "$ROOT/profiles/${TAG}_bench.json"
cp "$SRC/bench_c4_one_gpu.json" "$ROOT/profiles/${TAG}_bench_c4_one_gpu.json"
cp "$SRC/bench_c5_part.json" "$ROOT/profiles/${TAG}_bench_c5_part.json"
[ -f "$SRC/bench_c5_part_stream.json" ] && cp "$SRC/bench_c5_part_stream.json" "$ROOT/profiles/${TAG}_bench_c5_part_stream.json"
cp "$SRC"/stats_default/*/*_kernel_stats.csv "$ROOT/profiles/${TAG}_kernel_stats.csv"
for w in c2 streams glass_tree glass_stream s16_stream c5_tree c5_stream; do
    [ -d "$ROOT/gpurun_out/pmc_$w" ] || continue
    python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/pmc_$w" > "$ROOT/profiles/${TAG}_pmc_$w.json"
    cp "$ROOT"/gpurun_out/pmc_$w/stats/*/*_kernel_stats.csv "$ROOT/profiles/${TAG}_kernel_stats_$w.csv"
done
cp "$SRC/valu_rates.json" "$ROOT/profiles/${TAG}_valu_rates.json"
PTMI_PROFILE_BUILD_ID=$(cat "$SRC/build_id.txt" 2>/dev/null) python3 "$ROOT/tools/valu_roofline.py" "$ROOT/profiles/${TAG}_pmc_c2.json" "$ROOT/profiles/${TAG}_valu_rates.json" "$TAG" > /dev/null
PTMI_PROFILE_BUILD_ID=$(cat "$SRC/build_id.txt" 2>/dev/null) python3 "$ROOT/tools/valu_roofline.py" streams "$TAG"
python3 - "$ROOT" "$TAG" <<'PY'
import json, sys
root, tag = sys.argv[1], sys.argv[2]
d = json.load(open("%s/profiles/%s_pmc_c2.json" % (root, tag)))
k = next(v for n, v in d.items() if "render_inline_kernel" in n)
rd, wr = k["hbm_read_MB_per_call"] * 1e6, k["hbm_write_MB_per_call"] * 1e6
json.dump({"hbm_bytes_per_launch": round(rd + wr), "fetch_bytes_corrected": round(rd), "write_bytes": round(wr),
           "known_bytes_each_way": 7 * 1920 * 1080 * 4, "source": "profiles/%s_pmc_c2.json (FETCH_SIZE x 2 KiB, WRITE_SIZE KiB, separate passes)" % tag,
           "workload": "C2"}, open("%s/profiles/traffic.json" % root, "w"), indent=1)
PY
cp "$SRC/c4_part.json" "$ROOT/profiles/${TAG}_c4_part.json"
[ -s "$SRC/c5_part.json" ] && cp "$SRC/c5_part.json" "$ROOT/profiles/${TAG}_c5_part.json"
cp "$SRC/extra.json" "$ROOT/profiles/${TAG}_extra_measurements.json"
cp "$SRC/phase_stats.json" "$ROOT/profiles/${TAG}_phase_stats.json"
echo "profiles/${TAG}_* written"
for w in streams s16_stream; do [ -d "$ROOT/gpurun_out/pmcx_$w" ] && python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/pmcx_$w" > "$ROOT/profiles/${TAG}_pmc_issue_$w.json"; done
[ -s "$SRC/ab_options.txt" ] && cp "$SRC/ab_options.txt" "$ROOT/profiles/${TAG}_ab_options.txt"
for f in split_stats tail_stats tail_phases; do [ -s "$SRC/$f.json" ] && cp "$SRC/$f.json" "$ROOT/profiles/${TAG}_$f.json"; done
[ -s "$SRC/contracted.json" ] && cp "$SRC/contracted.json" "$ROOT/profiles/${TAG}_contracted_arithmetic.json"
[ -s "$SRC/traffic_terms.json" ] && cp "$SRC/traffic_terms.json" "$ROOT/profiles/${TAG}_traffic_terms.json"
[ -s "$SRC/soak_passes.json" ] && cp "$SRC/soak_passes.json" "$ROOT/profiles/${TAG}_soak_ordered_passes.json"
[ -s "$SRC/c0_calls.json" ] && cp "$SRC/c0_calls.json" "$ROOT/profiles/${TAG}_c0_calls.json"
for f in tree_stats; do [ -s "$SRC/$f.json" ] && cp "$SRC/$f.json" "$ROOT/profiles/${TAG}_$f.json"; done
true
