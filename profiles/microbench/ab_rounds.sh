#!/bin/bash
# Same-box A/B of two source trees: the committed tree of the previous round (git archive <commit> into abl/r5tree, built there)
# against this one; bench.py of each, interleaved, ROUNDS times.   gpurun -- 'bash profiles/microbench/ab_rounds.sh'
R=$GRAFT_REPO_ROOT
ROUNDS=${ROUNDS:-3}
ARGS=${ARGS:-"--no-cpu-baseline --no-extras --steps 20 --warmup 5"}
for i in $(seq $ROUNDS); do
  for t in abl/r5tree .; do
    (cd $R/$t && python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $i  tree %-12s %s  %.3f ms/step (sd %.2f)' % ('$t', d['config']['workload'][:10], d['ms_per_step'], d['ms_per_step_sd']))")
  done
done
