R=$GRAFT_REPO_ROOT
for v in base wg32ws; do echo "== $v"; CONV_BENCH_B=32 CONV_BENCH_ONLY="off" ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "off"; done
