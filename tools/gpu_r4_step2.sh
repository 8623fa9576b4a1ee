set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_streams.py tests/test_gpu_wavefront.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r4_tests2.log 2>&1 || true
tail -5 gpurun_out/r4_tests2.log
python tools/ab.py run --workloads glass_tree,glass_stream,glass_stream_uniform,glass_stream_g4,glass_stream_g8,glass_stream_g16,c5_tree,c5_stream,c5_stream_uniform,c5_stream_g8 default > gpurun_out/r4_ab2.txt 2>&1
cat gpurun_out/r4_ab2.txt
python tools/split_stats.py > gpurun_out/r4_split_stats2.json 2> gpurun_out/r4_split_stats2.err || tail -5 gpurun_out/r4_split_stats2.err
