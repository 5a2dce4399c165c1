cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_tmp.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_tmp.log | cut -c1-200
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_tmp/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms/step', tot/10e6)
for r in rows[:48]:
    n=r['Name'].replace('cnuda::(anonymous namespace)::','').replace('cnuda::','')[:70]
    print('%-70s x%-5s %8.2f ms/step %8.1f us avg %5.1f%%'%(n, int(r['Calls'])//10, float(r["TotalDurationNs"])/10e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
