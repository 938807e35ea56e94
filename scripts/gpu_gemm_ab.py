import ctypes, importlib, json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); p = capi.ptr
dev='cuda:0'
def st(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
res={}
for (M,N,K) in [(32768,768,256),(32768,512,512),(32768,256,512),(32768,256,256),(4096,256,256)]:
    A=torch.randn(M,K,device=dev); W=torch.randn(K,N); Wd=W.to(dev); C=torch.empty(M,N,device=dev); C6=torch.empty(M,N,device=dev)
    h=ctypes.c_void_p(); npad=ctypes.c_int()
    capi.check(lib, lib.dim_x3_create(p(W),K,N,ctypes.byref(h),ctypes.byref(npad)))
    t32=timeit(lambda: lib.dim_op_gemm_f32(p(A),K,p(Wd),N,0,None,None,0,p(C),N,M,N,K,0,st()))
    t6=timeit(lambda: lib.dim_op_gemm_x6_f32(p(A),K,h,npad.value,None,None,0,p(C6),N,M,N,K,0,st()))
    ref=(A[:2048].double()@Wd.double()); mag=(A[:2048].abs().double()@Wd.abs().double())
    e32=((C[:2048].double()-ref).abs()/mag).max().item(); e6=((C6[:2048].double()-ref).abs()/mag).max().item()
    fl=2.0*M*N*K
    res[f'{M}x{N}x{K}']={'fp32_TF':round(fl/t32/1e9,1),'x6_TF':round(fl/t6/1e9,1),'err_fp32':e32,'err_x6':e6}
    lib.dim_x3_destroy(h)
print(json.dumps(res,indent=1))
