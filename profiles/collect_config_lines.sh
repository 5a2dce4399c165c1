# bench lines of the BASELINE.json configs that are not the headline (configs[0], [1], [3], [4]) on one MI355X:
#   gpurun -- 'bash profiles/collect_config_lines.sh r2'   ->  gpurun_out/<tag>_bench_configs.jsonl  (copy to profiles/)
TAG=${1:-rX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_configs.jsonl
: > $OUT
for C in 0 1 3 4; do
  python3 $GRAFT_REPO_ROOT/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' >> $OUT
done
python3 - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    print('%-110s %8.2f img/s %8.2f ms/step' % (d['config']['workload'][:110], d['value'], d['ms_per_step']))
PY
