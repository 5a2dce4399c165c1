#!/bin/bash
# A/B of environment switches on ONE box in ONE call (boxes differ by +-1 %): every argument is an environment
# assignment (or "-" for none); the headline step is timed for each, interleaved, ROUNDS times.
#   bash profiles/ab_bench.sh - CNUDA_HCONV=0 CNUDA_HCONV=2
ROUNDS=${ROUNDS:-2}
R=$GRAFT_REPO_ROOT
for r in $(seq $ROUNDS); do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then v=$(python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_min'])")
    else v=$(env $kv python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_min'])"); fi
    echo "round $r  $kv  ms/step (mean, min): $v"
  done
done
