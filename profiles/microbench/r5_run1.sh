mkdir -p gpurun_out/r5
export CNUDA_DUMP_KERNELS=gpurun_out/r5/kernels_by_test.json
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > gpurun_out/r5/pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5/pytest1.log
tail -40 gpurun_out/r5/pytest1.log
unset CNUDA_DUMP_KERNELS
timeout 200 python profiles/microbench/dcn_offset_stats.py 3 > gpurun_out/r5/offset_stats.txt 2>&1
cat gpurun_out/r5/offset_stats.txt | tail -20
timeout 600 python bench.py > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err
echo "bench rc=$?"; head -c 600 gpurun_out/r5/bench1.json
