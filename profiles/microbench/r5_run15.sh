R=$GRAFT_REPO_ROOT
for v in gfull gnoscatter gnogather; do echo "== $v"; ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/dcn_layer.py --offsets small --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data"; done
