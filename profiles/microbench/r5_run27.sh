R=$GRAFT_REPO_ROOT
for v in pipe2 pipe3 pipe2 pipe3; do echo "== $v"; CONV_BENCH_B=32 CONV_BENCH_ONLY="stem,l0" ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "stem|l0"; done
ABL_LIB=$R/abl/lib_pipe3.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d_fwd_bwd or statistics" 2>&1 | tail -2
