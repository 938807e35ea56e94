import ctypes, importlib, json, sys, os, torch, torch.nn.functional as F
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
    x=torch.randn(Bn,H,H,cin,device=dev); w=(torch.randn(cout,cin,3,3)*0.05).contiguous(); b=torch.randn(cout,device=dev)
    wk=w.permute(2,3,1,0).contiguous().reshape(9,cin,cout).to(dev)
    Ho=H//2 if pool else H
    out=torch.empty(Bn,Ho,Ho,cout,device=dev); out6=torch.empty_like(out)
    h=ctypes.c_void_p(); capi.check(lib, lib.dim_convx6_create(p(w),cin,cout,ctypes.byref(h)))
    t32=timeit(lambda: lib.dim_op_conv3x3_nhwc_f32(p(x),p(wk),p(b),p(out),Bn,H,H,cin,cout,pool,1,st()))
    t6=timeit(lambda: lib.dim_op_conv3x3_x6_nhwc_f32(p(x),h,p(b),p(out6),Bn,H,H,cin,cout,pool,1,st()))
    fl=2.0*Bn*H*H*9*cin*cout
    # accuracy on one image crop vs fp64
    xs=x[:1,:64,:64].permute(0,3,1,2).double().cpu(); ref=torch.relu(F.conv2d(xs,w.double(),b.double().cpu(),padding=1))
    if pool: ref=F.max_pool2d(ref,2,2)
    mag=F.conv2d(xs.abs(),w.abs().double(),padding=1).max().item()
    c=(24 if pool else 48)
    e32=(out[0,:c,:c].permute(2,0,1).double().cpu()-ref[0,:,:c,:c]).abs().max().item()/mag
    e6=(out6[0,:c,:c].permute(2,0,1).double().cpu()-ref[0,:,:c,:c]).abs().max().item()/mag
    res[name]={'fp32_TF':round(fl/t32/1e9,1),'x6_TF':round(fl/t6/1e9,1),'err32':e32,'err6':e6}
    lib.dim_x3_destroy(h); del x,out,out6
print(json.dumps(res,indent=1))
