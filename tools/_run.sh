set -e
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout -k 10 300 python tools/small_latency.py > gpurun_out/r04_v1_small_latency.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python3 tools/batch_one.py > gpurun_out/r04_v1_config5_batch.log 2>&1
cp $(find gpurun_out/prof_c5 -name "c5_kernel_stats.csv" | head -1) gpurun_out/r04_v1_config5_batch_kernel_stats.csv
timeout -k 10 900 python bench.py --steps 50 --warmup 5 > gpurun_out/r04_v1_bench.json 2> gpurun_out/r04_v1_bench.err
tail -c 300 gpurun_out/r04_v1_bench.json
