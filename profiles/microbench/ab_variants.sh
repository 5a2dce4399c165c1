#!/bin/bash
# bash profiles/microbench/ab_variants.sh base v1 v2 ...   (abl/lib_<name>.so, built by build_variant.sh)
# per variant: the convolution parity tests, the per-layer table; then the headline step of all of them, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/ab; mkdir -p $O
TESTS=${TESTS:-"tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py"}
KEXPR=${KEXPR:-"conv or 1x1 or halo"}
for v in "$@"; do
  [ "$v" = base ] && continue
  ABL_LIB=$R/abl/lib_$v.so timeout 900 python -m pytest $TESTS -q -x -m gpu -k "$KEXPR" -p no:cacheprovider > $O/tests_$v.log 2>&1
  echo "$v tests rc=$? $(tail -1 $O/tests_$v.log)"
done
for v in "$@"; do
  ABL_LIB=$R/abl/lib_$v.so timeout 300 python profiles/conv_layers.py --iters 3 > $O/layers_$v.txt 2>&1
done
VARIANTS="$*" python - <<'PY'
import os,re
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/ab'
vs=os.environ['VARIANTS'].split()
rows={}
for v in vs:
    for l in open('%s/layers_%s.txt'%(O,v)):
        m=re.match(r'(\S+ \S+ \S+)\s+(\S.*?)\s+([\d.]+)\s+([\d.]+)\s*$',l)
        if m: rows.setdefault((m.group(1),m.group(2)[:44]),{})[v]=float(m.group(3))
tot={v:0.0 for v in vs}
for k,d in rows.items():
    print('%-30s %-44s '%k+' '.join('%s %7.1f'%(v,d.get(v,float('nan'))) for v in vs))
    for v in vs: tot[v]+=d.get(v,0.0)
print('sum (us): '+' '.join('%s %.0f'%(v,tot[v]) for v in vs))
PY
libs=""; for v in "$@"; do libs="$libs abl/lib_$v.so"; done
ROUNDS=${ROUNDS:-2} STEPS=12 bash profiles/microbench/ab_lib.sh $libs
