for off in small zero; do echo "== offsets=$off"; CNUDA_DCNQ=1 python3 profiles/dcn_layer.py --time --iters 3 --offsets $off 2>/dev/null | grep -E "^B=|dcnq"; done
python -m pytest tests/test_gpu_dcn.py tests/test_gpu_fullsize.py -m gpu -q -k "quad" 2>&1 | tail -3
