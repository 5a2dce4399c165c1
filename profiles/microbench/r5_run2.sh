R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
for v in st3 st3occ3 occ3; do
  ABL_LIB=$R/abl/lib_$v.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x -m gpu -k "conv or 1x1 or halo" -p no:cacheprovider > $O/run2_tests_$v.log 2>&1
  echo "$v tests rc=$? $(tail -1 $O/run2_tests_$v.log)"
done
for v in base st3 occ3 st3occ3; do
  ABL_LIB=$R/abl/lib_$v.so timeout 300 python profiles/conv_layers.py --iters 3 > $O/run2_layers_$v.txt 2>&1
done
python - <<'PY'
import os,re
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5'
rows={}
for v in ('base','st3','occ3','st3occ3'):
    for l in open('%s/run2_layers_%s.txt'%(O,v)):
        m=re.match(r'(\S+ \S+ \S+)\s+(\S.*?)\s+([\d.]+)\s+([\d.]+)\s*$',l)
        if m: rows.setdefault((m.group(1),m.group(2)[:44]),{})[v]=float(m.group(3))
for k,d in rows.items():
    print('%-30s %-44s '%k+' '.join('%s %7.1f'%(v,d.get(v,float('nan'))) for v in ('base','st3','occ3','st3occ3')))
PY
ROUNDS=2 STEPS=12 bash profiles/microbench/ab_lib.sh abl/lib_base.so abl/lib_st3.so abl/lib_occ3.so abl/lib_st3occ3.so
