R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
for sg in 1 2; do
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --dcn-offset-std $sg > $O/run6_sigma$sg.json 2>$O/run6.err
SG=$sg python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/run6_sigma%s.json'%os.environ['SG']).readline())
print('sigma', os.environ['SG'], d['ms_per_step'], d['ms_per_step_min'])
r=d['roofline']
for k,v in list(r['all_mfma_kernels'].items()): 
    if 'dcn' in k.lower() or 'Dcn' in k or 'shortk' in k: print('%8.3f ms %7.2f TF %4d  %s'%(v['ms_per_step'],v['tflops'],v['launches'],k))
for k,v in r['hbm_kernels'].items(): print('%8.3f ms %7.1f GB/s %4d  %s'%(v['ms_per_step'],v['gb_per_s'],v['launches'],k))
PY
done
