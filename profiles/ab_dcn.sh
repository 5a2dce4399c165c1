#!/bin/bash
# DCN layer timings for environment switches on ONE box: bash profiles/ab_dcn.sh - CNUDA_DCNW=0
R=$GRAFT_REPO_ROOT
for kv in "$@"; do
  for off in small zero; do
    echo "== $kv offsets=$off"
    if [ "$kv" = "-" ]; then python3 $R/profiles/dcn_layer.py --time --iters 3 --offsets $off 2>/dev/null | grep -v "^/opt"
    else env $kv python3 $R/profiles/dcn_layer.py --time --iters 3 --offsets $off 2>/dev/null | grep -v "^/opt"; fi
  done
done
