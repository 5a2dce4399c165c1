# keep the sampled columns (default) vs re-sample in the weight gradient: DCN layer times and the whole step, one box
for k in 1 0; do echo "== CNUDA_DCN_KEEP_COLS=$k"; CNUDA_DCN_KEEP_COLS=$k python3 profiles/dcn_layer.py --time --iters 3 --offsets zero 2>/dev/null | grep -E "^B=|dcnw|DcnW|DcnColW"; done
for k in 1 0 1 0; do CNUDA_DCN_KEEP_COLS=$k python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KEEP_COLS=$k', d['ms_per_step'], d['value'])"; done
