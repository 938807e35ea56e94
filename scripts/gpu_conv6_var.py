import ctypes, importlib, json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); p = capi.ptr
dev='cuda:0'
def st(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
Bn=16
shapes={'conv1b':(64,64,1024,1),'conv2a':(64,64,512,0),'conv3b':(128,128,256,1),'conv4a':(128,128,128,0),'convPa':(128,256,128,0)}
res={}
for name,(cin,cout,H,pool) in shapes.items():
    x=torch.randn(Bn,H,H,cin,device=dev); w=(torch.randn(cout,cin,3,3)*0.05).contiguous(); b=torch.randn(cout,device=dev)
    Ho=H//2 if pool else H
    out=torch.empty(Bn,Ho,Ho,cout,device=dev); ref=None
    h=ctypes.c_void_p(); capi.check(lib, lib.dim_convx6_create(p(w),cin,cout,ctypes.byref(h)))
    fl=2.0*Bn*H*H*9*cin*cout
    for rnd in range(2):
        for v in (0,1,2):
            lib.dim_tune_set(2,v)
            ms=timeit(lambda: lib.dim_op_conv3x3_x6_nhwc_f32(p(x),h,p(b),p(out),Bn,H,H,cin,cout,pool,1,st()))
            res.setdefault(name,{}).setdefault(v,[]).append(round(fl/ms/1e9,1))
            if ref is None: ref=out.clone()
            else: assert torch.equal(ref,out),(name,v)
    lib.dim_x3_destroy(h); del x,out,ref
print(json.dumps(res))
