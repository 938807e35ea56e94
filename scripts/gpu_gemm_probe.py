"""What bounds the wide fp16x3 GEMM blocks (gemm_x6.hip)?  Times LightGlue's ffn.3 shape (204800 x 512 -> 256, + residual;
gemm_x6_kernel<2,128,2,4>) and the fused ffn.0 + LayerNorm + GELU shape (204800 x 512 -> 512; gemm_x6_ffn_ln_kernel) under the
PROBE instantiations (dim_tune_set key 13): 0 product; 1 activations cache-resident; 2 weight fragments always chunk 0 (L1 hits);
4 no MFMAs; 8 weight fragments loaded once; 9 = 1 + 8.  Results of probes != 0 are wrong by design.  us per launch, HIP events."""
import ctypes, importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(str(capi.LIB_PATH.parent / 'libdim_hip_research.so')); capi.install(lib, None)   # research build: dim_tune_set keys 12-15 (timing probes / prototypes) exist only there
p = lambda t: ctypes.c_void_p(t.data_ptr())
M, K = 204800, 512
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g)).cuda(); R = torch.randn(M, 256, generator=g).cuda()
def handle(N):
    W = (torch.randn(K, N, generator=g) / K ** 0.5).contiguous()
    h, npad = ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W), K, N, ctypes.byref(h), ctypes.byref(npad)) == 0
    return h, npad.value
h256, np256 = handle(256); h512, _ = handle(512)
b256, b512 = torch.zeros(256).cuda(), torch.zeros(512).cuda(); gm, bt = torch.ones(512).cuda(), torch.zeros(512).cuda()
C256, C512 = torch.empty(M, 256).cuda(), torch.empty(M, 512).cuda()
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
keep = {}
for probe in [int(x) for x in (sys.argv[1:] or "0 1 2 4 8 9 0".split())]:
    lib.dim_tune_set(13, probe)
    a = t(lambda: capi.check(lib, lib.dim_op_gemm_x6_f32(p(A), K, h256, np256, p(b256), p(R), 256, p(C256), 256, M, 256, K, 0, stream)))
    b = t(lambda: capi.check(lib, lib.dim_op_gemm_x6_ln_gelu_f32(p(A), K, h512, p(b512), p(gm), p(bt), p(C512), 512, M, K, stream)))
    res[f"probe{probe}" + ("_again" if f"probe{probe}" in res else "")] = {"ffn3_us": a, "ffn0_ln_us": b}
    if probe in (0, 100):   # 100 = the pipelined K loop: a product candidate, results must equal probe 0 bit for bit
        torch.cuda.synchronize()
        if probe in keep: continue
        keep[probe] = (C256.clone(), C512.clone())
lib.dim_tune_set(13, 0)
if 0 in keep and 100 in keep:
    res["pipe_equals_product"] = [bool(torch.equal(keep[0][0], keep[100][0])), bool(torch.equal(keep[0][1], keep[100][1]))]
print(json.dumps(res))
