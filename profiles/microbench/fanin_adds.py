import os, sys, collections
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/centernet-uda_amd')
import torch, bench
import hip_runtime.fanout as fo
import traceback
log=collections.Counter()
orig=fo.add_into
def add_into(slot, g):
    before = slot.buf
    r = orig(slot, g)
    if before is not None and not any(t.data_ptr()==g.data_ptr() for t in slot.included) and before.data_ptr()!=g.data_ptr():
        log[(tuple(g.shape), 'inplace' if r.data_ptr()==before.data_ptr() else 'new')] += 1
    return r
fo.add_into = add_into
dev=torch.device('cuda',0); torch.cuda.set_device(dev)
plugin=bench.build_plugin(dev, parallel=False)
batch=bench.synthetic_batch(16,512,42,dev)
for _ in range(2): plugin.step(bench.fresh(batch))
log.clear()
plugin.step(bench.fresh(batch)); torch.cuda.synchronize()
for k,v in sorted(log.items(), key=lambda kv:-kv[1]): print(v,k)
