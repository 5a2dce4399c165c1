R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
for v in base nogather noscatter; do for o in small sigma1; do
  echo "== $v offsets=$o"; ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/dcn_layer.py --offsets $o --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data|dcnw_fwd|shortk|DcnColW"
done; done > $O/dcn_roles.txt 2>&1
cat $O/dcn_roles.txt
bash profiles/microbench/ab_variants.sh base rs2 stag rs2stag
