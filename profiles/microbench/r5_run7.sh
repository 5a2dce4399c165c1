R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dcn.py tests/test_gpu_dla.py tests/test_gpu_batched_domains.py tests/test_gpu_mobilenetv2.py tests/test_gpu_export.py -q -x -m gpu -p no:cacheprovider > $O/run7_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run7_tests.log)"; grep -E "^E " $O/run7_tests.log | head -20
python profiles/microbench/ab_bn_stats.py 2>/dev/null
