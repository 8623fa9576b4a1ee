set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests1.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_tests1.log
tail -3 gpurun_out/r4_tests1.log
python tools/split_stats.py > gpurun_out/r4_split_stats_base.json 2> gpurun_out/r4_split_stats_base.err
python tools/ab.py run --workloads glass_tree,glass_stream,streams,s16_stream > gpurun_out/r4_ab_base.txt 2>&1
cat gpurun_out/r4_ab_base.txt
python bench.py > gpurun_out/r4_bench_base.json 2> gpurun_out/r4_bench_base.err
tail -c 600 gpurun_out/r4_bench_base.json
