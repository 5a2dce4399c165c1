R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
CNUDA_DUMP_KERNELS=$O/k3.json timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -p no:cacheprovider -k "batchnorm_statistics" > $O/run16_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/run16_tests.log)"; grep -E "^E " $O/run16_tests.log | head
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/k3.json'))
for t,ks in d.items():
    print(t.split('::')[-1], [k.split('(')[0].replace('void cnuda::','')[:60] for k in ks if 'igemm_fwd' in k])
PY
