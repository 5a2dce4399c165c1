#!/bin/bash
# A/B of library builds on ONE box in ONE call: every argument is a path to a build of libcenternet_uda_hip.so
# (relative to the repo root); the headline step is timed for each, interleaved, ROUNDS times.
#   bash profiles/microbench/ab_lib.sh gpurun_out/abl/lib_old.so gpurun_out/abl/lib_new.so
ROUNDS=${ROUNDS:-2}
R=$GRAFT_REPO_ROOT
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    v=$(ABL_LIB=$R/$lib python3 $R/profiles/microbench/bench_lib.py --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_min'])")
    echo "round $r  $lib  ms/step (mean, min): $v"
  done
done
