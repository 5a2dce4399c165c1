R=$GRAFT_REPO_ROOT
bash profiles/microbench/r5_full_tests.sh
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/r5/mid_bench.json 2> $R/gpurun_out/r5/mid_bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/mid_bench.json').readline())
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])
for k,v in list(d['roofline']['all_mfma_kernels'].items())[:40]:
    print('  %7.3f ms %7.2f TF %3d  %s'%(v['ms_per_step'],v['tflops'],v['launches'],k))
PY
