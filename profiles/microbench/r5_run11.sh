R=$GRAFT_REPO_ROOT
for v in base sc3; do echo "== $v"; ABL_LIB=$R/abl/lib_$v.so timeout 300 python profiles/conv_layers.py --iters 3 2>&1 | grep -E "512/s|256/s" ; done
