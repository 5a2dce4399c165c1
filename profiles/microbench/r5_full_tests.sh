R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
export CNUDA_DUMP_KERNELS=$O/kernels_by_test2.json
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $O/full_tests.log 2>&1
echo "pytest rc=$?"; tail -25 $O/full_tests.log; grep -E "^E  " $O/full_tests.log | head -30
