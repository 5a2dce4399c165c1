R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for v in base c8new; do echo "== $v"; CONV_BENCH_B=32 CONV_BENCH_ONLY="l1,l2 32" ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "^l"; done
