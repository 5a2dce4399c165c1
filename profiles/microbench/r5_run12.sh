R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fanout.py tests/test_gpu_dla.py -q -x -m gpu -p no:cacheprovider > $O/run12_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run12_tests.log)"; grep -E "^E " $O/run12_tests.log | head -20
timeout 300 python profiles/conv_layers.py --iters 3 2>&1 | grep -E "512/s|256/s"
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_min'])"
