R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -x -m gpu -p no:cacheprovider -k "dcn or Dcn or dcn_layer" > $O/run14_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run14_tests.log)"; grep -E "^E " $O/run14_tests.log | head -20
for off in small sigma1; do echo "== $off"; timeout 120 python profiles/dcn_layer.py --offsets $off --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data"; done
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_min']); print({k:v for k,v in d['roofline']['hbm_kernels'].items()})"
