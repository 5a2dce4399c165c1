R=$GRAFT_REPO_ROOT
for v in colsws1 samp2; do echo "== $v"; DCN_LAYER_SHAPES=small_maps ABL_LIB=$R/abl/lib_$v.so timeout 200 python profiles/dcn_layer.py --offsets small --iters 3 --time 2>&1 | grep -E "B=|sample"; done
timeout 600 python -m pytest tests/test_gpu_dcn.py -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "dcn_layer and (256to or 512to)" 2>&1 | tail -2
