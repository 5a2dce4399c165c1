cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections,json
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
res={}
for C in ('FETCH_SIZE','WRITE_SIZE'):
    f=glob.glob(root+'/pmc_%s/*/*counter_collection.csv'%C)[0]
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']!=C: continue
        n=r['Kernel_Name'].replace('cnuda::(anonymous namespace)::','').replace('cnuda::','').split('(')[0].replace('void ','').replace(', false>','>')
        agg[n][0]+=float(r['Counter_Value']); agg[n][1]+=1
    res[C]=agg
names=sorted(res['FETCH_SIZE'], key=lambda n:-res['FETCH_SIZE'][n][0])[:40]
out={}
for n in names:
    f,c=res['FETCH_SIZE'][n]; w,c2=res['WRITE_SIZE'].get(n,[0,1])
    out[n]={'launches':c,'fetch_kb_per_launch':f/c,'write_kb_per_launch':w/max(c2,1)}
    print('%-60s x%-4d FETCH %10.1f MB/launch (raw)  WRITE %10.1f MB/launch'%(n[:60],c,f/c/1024,w/max(c2,1)/1024))
# every kernel of the trace (not only the 40 listed): bytes of ONE step = total / the 2 steps the command runs (warm-up + timed)
STEPS=2
total=sum(v[0] for v in res['FETCH_SIZE'].values())+sum(v[0] for v in res['WRITE_SIZE'].values())
print('all kernels: %.1f GB per step (FETCH_SIZE as reported + WRITE_SIZE, %d steps traced)'%(total*1024/STEPS/1e9,STEPS))
import sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']+'/profiles')
import fingerprint
out['_meta']={'steps_traced':STEPS,'hbm_bytes_per_step':int(total*1024/STEPS),'code_sha16':fingerprint.code_fingerprint(),'command':'bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0',
              'units':'KB per launch, averaged over the launches of one step; FETCH_SIZE as reported (the guide: double it for wide coalesced reads; these kernels gather 4-8 bytes per lane, uncalibrated)'}
json.dump(out, open(root+'/pmc_traffic.json','w'), indent=1)
PY
