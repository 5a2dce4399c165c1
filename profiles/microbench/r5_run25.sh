R=$GRAFT_REPO_ROOT
for v in 0 1 2 3; do echo "== HW_ABL=$v"; CONV_BENCH_B=32 CONV_BENCH_ONLY="off" ABL_LIB=$R/abl/lib_hwabl$v.so timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "^off"; done
