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
# launches per STEP: the trace above also holds the one-time work of the process (arena / optimizer-state copies: ~640
# copyBuffer launches), so total calls / steps overstates a step.  A second, shorter trace of the same command differs from
# the first only in the number of steps: (calls of 10 steps - calls of 5 steps) / 5 is the step's own count.
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp5
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_tmp5.log 2>&1
python3 - <<'PY'
import csv,glob,json,os
R=os.environ['GRAFT_REPO_ROOT']
def calls(d):
    f=glob.glob(R+'/gpurun_out/%s/*/*kernel_stats.csv'%d)[0]
    return {r['Name']:int(r['Calls']) for r in csv.DictReader(open(f))}
a,b=calls('prof_tmp'),calls('prof_tmp5')
per={k:(a[k]-b.get(k,0))/5.0 for k in a}
tot=sum(per.values())
json.dump({'launches_per_step':tot,'how':'(kernel calls of a 10-step trace - calls of a 5-step trace) / 5','one_time_launches':sum(a.values())-10*tot,
           'by_kernel':{k:v for k,v in sorted(per.items(),key=lambda kv:-kv[1]) if v}}, open(R+'/gpurun_out/launches_per_step.json','w'), indent=1)
print('launches per step (differential):', tot, ' one-time launches of the process:', sum(a.values())-10*tot)
PY
