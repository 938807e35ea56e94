import ctypes, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); p = capi.ptr
dev='cuda:0'
M,N,K=32768,512,512
A=torch.randn(M,K,device=dev); W=torch.randn(K,N); Wd=W.to(dev); C=torch.empty(M,N,device=dev)
h=ctypes.c_void_p(); npad=ctypes.c_int()
lib.dim_x3_create(p(W),K,N,ctypes.byref(h),ctypes.byref(npad))
for _ in range(3):
    lib.dim_op_gemm_f32(p(A),K,p(Wd),N,0,None,None,0,p(C),N,M,N,K,0,None)
    lib.dim_op_gemm_x6_f32(p(A),K,h,npad.value,None,None,0,p(C),N,M,N,K,0,None)
torch.cuda.synchronize()
