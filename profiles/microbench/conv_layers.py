"""Per-layer forward / input-gradient / weight-gradient time of the DLA-34 convolution shapes (B = 16 unless CONV_BENCH_B),
through the C ABI, timed by the library's own events (cnuda_prof_*).  ABL_LIB=<path to another build of the .so> times that
build instead (A/B of kernel changes on ONE box: boxes differ by +-1 %)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'centernet-uda_amd'))
import hip_runtime as hr
if os.environ.get("ABL_LIB"): hr.LIB_PATH = os.environ["ABL_LIB"]
from hip_runtime import ops
dev='cuda:0'
B=int(os.environ.get("CONV_BENCH_B", 16))
shapes=[ # name, C, H, Co, k, s
 ('stem7x7',3,512,16,7,1),('l0 16-16',16,512,16,3,1),('l1 16-32s2',16,512,32,3,2),
 ('l2 32-64s2',32,256,64,3,2),('l2 64-64',64,128,64,3,1),('l2 root128-64',128,128,64,1,1),
 ('l3 64-128s2',64,128,128,3,2),('l3 128-128',128,64,128,3,1),('l3 root448-128',448,64,128,1,1),
 ('l4 128-256s2',128,64,256,3,2),('l4 256-256',256,32,256,3,1),('l4 root896-256',896,32,256,1,1),
 ('l5 256-512s2',256,32,512,3,2),('l5 512-512',512,16,512,3,1),
 ('off 64-27',64,128,27,3,1),('off 128-27',128,64,27,3,1),
 ('head 64-256',64,128,256,3,1),('head 256-6',256,128,6,1,1),
]
def timeit(kind, fn, n=3):
    for _ in range(2): fn()
    hr.prof_begin(64)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    r=hr.prof_end()
    (name,d),=r.items()
    return d['ms']/d['launches'], d['flops']/d['launches'], name
print('%-18s %28s %28s %28s'%('layer','fwd ms / TF','dgrad ms / TF','wgrad ms / TF'))
ONLY=os.environ.get('CONV_BENCH_ONLY')
for name,C,H,Co,k,s in shapes:
    if ONLY and not any(name.startswith(o) for o in ONLY.split(',')): continue
    x=torch.randn(B,C,H,H,device=dev); w=torch.randn(Co,C,k,k,device=dev)*0.05
    x.requires_grad_(True); w.requires_grad_(True)
    y=ops.conv2d(x,w,None,s,k//2); gy=torch.randn_like(y)
    L=hr.lib(); g=(B,C,H,H,Co,k,k,s,s,k//2,k//2)
    ws=hr.workspace(L.cnuda_conv2d_workspace_bytes(*g), x.device)
    gx=torch.empty_like(x); gw=torch.empty_like(w)
    Ho=y.shape[2]
    def f():
        hr.prof_arm('conv_fwd',B,C,H,H,Co,k,k,Ho,Ho); hr.check(L.cnuda_conv2d_forward(hr.ptr(x),hr.ptr(w),None,hr.ptr(y),*g,-1.0,hr.ptr(ws),ws.numel(),hr.stream()))
    def d():
        hr.prof_arm('conv_dgrad',B,C,H,H,Co,k,k,Ho,Ho); hr.check(L.cnuda_conv2d_backward_data(hr.ptr(gy),hr.ptr(w),hr.ptr(gx),*g,hr.ptr(ws),ws.numel(),hr.stream()))
    def wg():
        hr.prof_arm('conv_wgrad',B,C,H,H,Co,k,k,Ho,Ho); hr.check(L.cnuda_conv2d_backward_weight(hr.ptr(x),hr.ptr(gy),hr.ptr(gw),None,*g,hr.ptr(ws),ws.numel(),hr.stream()))
    out=[]
    for fn in (f,d,wg):
        ms,fl,kn=timeit('',fn); out.append('%7.3f ms %6.1f TF'%(ms, fl/ms/1e9))
    print('%-18s %28s %28s %28s'%(name,*out))
