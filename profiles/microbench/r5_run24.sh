R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dcn.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "dcn_layer" 2>&1 | tail -3
for v in gbuf0 gbuf1; do echo "== $v"; ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/dcn_layer.py --offsets small --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data"; done
for v in gbuf0 gbuf1; do echo "== $v sigma1"; ABL_LIB=$R/abl/lib_$v.so timeout 120 python profiles/dcn_layer.py --offsets sigma1 --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data"; done
