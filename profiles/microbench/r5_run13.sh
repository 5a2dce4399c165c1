R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_dla.py tests/test_gpu_resnet.py tests/test_gpu_mobilenetv2.py tests/test_gpu_batched_domains.py tests/test_gpu_matrix_mode.py -q -x -m gpu -p no:cacheprovider > $O/run13_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run13_tests.log)"; grep -E "^E " $O/run13_tests.log | head -20
timeout 300 python profiles/conv_layers.py --iters 3 2>&1 | grep -E "512/s"
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $O/run13_bench.json 2>$O/run13_bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/run13_bench.json').readline())
print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['kernel'], d['roofline']['frac'])
r=d['roofline']
for k,v in list(r['all_mfma_kernels'].items())[:14]: print('%8.3f ms %7.2f TF %4d  %s'%(v['ms_per_step'],v['tflops'],v['launches'],k))
PY
python profiles/microbench/ab_bn_stats.py 2>/dev/null
