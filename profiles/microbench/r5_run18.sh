R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 5; do echo "== SC_ABL=$v"; CONV_BENCH_B=32 CONV_BENCH_ONLY="stem,l0" ABL_LIB=$R/abl/lib_scabl$v.so timeout 120 python profiles/microbench/conv_layers.py 2>&1 | grep -E "stem|l0"; done
