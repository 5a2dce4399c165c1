R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dla.py tests/test_gpu_fullsize.py -q -x -m gpu -p no:cacheprovider -k "census or dla or step or dcn_layer" > $O/run9_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run9_tests.log)"; grep -E "^E " $O/run9_tests.log | head -20
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run9_bench.json 2>$O/run9_bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/run9_bench.json').readline())
print(d['ms_per_step'], d['ms_per_step_min']); print(d['dcn_offsets']); print(d['inference']); print({k:v['ms_per_step'] for k,v in d['other_configs'].items()})
PY
