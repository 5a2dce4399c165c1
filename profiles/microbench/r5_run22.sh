R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d_fwd_bwd" 2>&1 | tail -5
echo "== new (hwgrad)"; CONV_BENCH_B=32 CONV_BENCH_ONLY="off" timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "off"
echo "== CNUDA_HWGRAD=0"; CNUDA_HWGRAD=0 CONV_BENCH_B=32 CONV_BENCH_ONLY="off" timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "off"
