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
shapes={'conv1b':(64,64,1024,1),'conv2a':(64,64,512,0),'conv3a':(64,128,256,0),'conv3b':(128,128,256,1),'conv4a':(128,128,128,0),'convPa':(128,256,128,0)}
res={}
for name,(cin,cout,H,pool) in shapes.items():
    x=torch.randn(Bn,H,H,cin,device=dev); w=torch.randn(9,cin,cout,device=dev)*0.05; b=torch.randn(cout,device=dev)
    Ho=H//2 if pool else H
    out=torch.empty(Bn,Ho,Ho,cout,device=dev); ref=None
    for rnd in range(2):
      for v in [int(a) for a in sys.argv[1:]] or [0,1,2,3]:
        lib.dim_tune_set(0,v)
        ms=timeit(lambda: lib.dim_op_conv3x3_nhwc_f32(p(x),p(w),p(b),p(out),Bn,H,H,cin,cout,pool,1,st()))
        fl=2.0*Bn*H*H*9*cin*cout
        res.setdefault(name,{}).setdefault(v,[]).append(round(fl/ms/1e9,1))
        if ref is None: ref=out.clone()
        else: assert (ref-out).abs().max().item() < 1e-3*ref.abs().max().item(), (name,v)
    del x,out,ref
print(json.dumps(res))
