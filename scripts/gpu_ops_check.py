"""GPU check + micro-benchmark of the operator-level kernels (run on the MI355X box)."""
import ctypes, importlib, json, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
capi = importlib.import_module('deep-image-matching_amd.capi')
lib = capi.load(); p = capi.ptr
dev = 'cuda:0'
torch.manual_seed(0)
res = {}
def st(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M,N,K,bt) in [(200,65,64,0),(2048,768,256,0),(2048,2048,256,1)]:
    A=torch.randn(M,K); B=torch.randn(N,K) if bt else torch.randn(K,((N+3)//4)*4); bias=torch.randn(N); R=torch.randn(M,N)
    Ad,Bd,bd,Rd=[t.to(dev) for t in (A,B,bias,R)]; C=torch.zeros(M,N,device=dev)
    capi.check(lib, lib.dim_op_gemm_f32(p(Ad),K,p(Bd),B.shape[1],bt,p(bd),p(Rd),N,p(C),N,M,N,K,1,st()))
    ref=torch.relu((A@B.T if bt else A@B[:,:N])+bias+R)
    res[f'gemm_{M}_{N}_{K}_{bt}_err']=(C.cpu()-ref).abs().max().item()
for (cin,cout,H,W,pool) in [(64,64,20,37,1),(64,128,9,33,0),(128,256,17,16,0),(128,128,16,64,1)]:
    x=torch.randn(2,cin,H,W); w=torch.randn(cout,cin,3,3)*0.1; b=torch.randn(cout)
    xin=x.permute(0,2,3,1).contiguous().to(dev); wk=w.permute(2,3,1,0).contiguous().reshape(9,cin,cout).to(dev); bd=b.to(dev)
    Ho,Wo=(H//2,W//2) if pool else (H,W)
    out=torch.full((2,Ho,Wo,cout),-7.0,device=dev)
    capi.check(lib, lib.dim_op_conv3x3_nhwc_f32(p(xin),p(wk),p(bd),p(out),2,H,W,cin,cout,pool,1,st()))
    ref=torch.relu(F.conv2d(x,w,b,padding=1))
    if pool: ref=F.max_pool2d(ref,2,2)
    res[f'conv_{cin}_{cout}_{H}_{W}_{pool}_err']=(out.cpu()-ref.permute(0,2,3,1)).abs().max().item()
# ---- timing ----
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
Bn=4
for name,(cin,cout,H,pool) in {'conv1b':(64,64,1024,1),'conv2a':(64,64,512,0),'conv3a':(64,128,256,0),'conv3b':(128,128,256,1),'conv4a':(128,128,128,0),'convPa':(128,256,128,0)}.items():
    x=torch.randn(Bn,H,H,cin,device=dev); w=torch.randn(9,cin,cout,device=dev)*0.05; b=torch.randn(cout,device=dev)
    Ho=H//2 if pool else H
    out=torch.empty(Bn,Ho,Ho,cout,device=dev)
    ms=timeit(lambda: lib.dim_op_conv3x3_nhwc_f32(p(x),p(w),p(b),p(out),Bn,H,H,cin,cout,pool,1,st()))
    fl=2.0*Bn*H*H*9*cin*cout
    res[name+'_ms']=ms; res[name+'_TF']=fl/ms/1e9
    del x,out
x=torch.rand(Bn,1024,1024,device=dev); w=torch.randn(9,64,device=dev); b=torch.randn(64,device=dev); out=torch.empty(Bn,1024,1024,64,device=dev)
res['conv1a_ms']=timeit(lambda: lib.dim_op_conv1a_f32(p(x),p(w),p(b),p(out),Bn,1024,1024,st()))
res['conv1a_GBs']=Bn*1024*1024*65*4/res['conv1a_ms']/1e6
del out
for (M,N,K,bt) in [(32768,768,256,0),(32768,512,512,0),(32768,256,512,0),(2048,2048,256,1)]:
    A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev) if bt else torch.randn(K,N,device=dev); C=torch.empty(M,N,device=dev)
    ms=timeit(lambda: lib.dim_op_gemm_f32(p(A),K,p(B),B.shape[1],bt,None,None,0,p(C),N,M,N,K,0,st()))
    res[f'gemm_{M}_{N}_{K}_{bt}_ms']=ms; res[f'gemm_{M}_{N}_{K}_{bt}_TF']=2.0*M*N*K/ms/1e9
print(json.dumps(res, indent=1))
import os; os.makedirs('gpurun_out',exist_ok=True); json.dump(res, open('gpurun_out/ops_check.json','w'), indent=1)
