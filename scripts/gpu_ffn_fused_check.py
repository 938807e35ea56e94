"""GPU: the fused feed-forward kernel (ffn.0 -> LayerNorm -> GELU -> ffn.3 + residual, gemm_x6_ffn_fused_kernel) vs fp64 at sizes that
put two workgroups on every CU, twice (determinism), and its time at the bench shape against the two kernels it replaces."""
import ctypes, importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_ops_emu import _ffn_fused_case
capi = importlib.import_module('deep-image-matching_amd.capi')
# DIM_LIB=<path of a variant library> DIM_TUNE14=<value>: check / time a research variant of the kernel (e.g. the 128-row block: 14 = 128)
lib = capi.load(os.environ["DIM_LIB"]) if os.environ.get("DIM_LIB") else capi.load()
if os.environ.get("DIM_LIB"):
    capi.install(lib, None)
if os.environ.get("DIM_TUNE14"):
    assert lib.dim_tune_set(14, int(os.environ["DIM_TUNE14"])) == 0
p = lambda t: ctypes.c_void_p(t.data_ptr())
res = {}
for M in (2048 + 37, 65536, 204800):
    C1, ref = _ffn_fused_case(lib, M, 512, seed=M, device="cuda")
    C2, _ = _ffn_fused_case(lib, M, 512, seed=M, device="cuda")
    err = (C1.double() - ref).abs().max(1).values
    res[str(M)] = {"max_err": float(err.max()), "bad_rows": int((err > 2e-5).sum()), "bit_equal_rerun": bool(torch.equal(C1, C2)),
                   "first_bad_rows": (err > 2e-5).nonzero().reshape(-1)[:8].tolist()}
# timing at the bench shape
M, K = 204800, 512
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).cuda(); R = torch.randn(M, 256, generator=g).cuda()
def handle(Kk, N, perm=False):
    W = (torch.randn(Kk, N, generator=g) / Kk ** 0.5).contiguous()
    h, npad = ctypes.c_void_p(), ctypes.c_int()
    assert (lib.dim_x3_create_kperm if perm else lib.dim_x3_create)(p(W), Kk, N, ctypes.byref(h), ctypes.byref(npad)) == 0
    return h, npad.value
h0, _ = handle(512, 512); h3, np3 = handle(512, 256); h3p, _ = handle(512, 256, True)
b256, b512 = torch.zeros(256).cuda(), torch.zeros(512).cuda(); gm, bt = torch.ones(512).cuda(), torch.zeros(512).cuda()
C256, C512 = torch.empty(M, 256).cuda(), torch.empty(M, 512).cuda()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)
def two():
    capi.check(lib, lib.dim_op_gemm_x6_ln_gelu_f32(p(A), K, h0, p(b512), p(gm), p(bt), p(C512), 512, M, K, stream))
    capi.check(lib, lib.dim_op_gemm_x6_f32(p(C512), 512, h3, np3, p(b256), p(R), 256, p(C256), 256, M, 256, 512, 0, stream))
def one():
    capi.check(lib, lib.dim_op_ffn_fused_f32(p(A), K, h0, p(b512), p(gm), p(bt), h3p, p(b256), p(R), 256, p(C256), 256, M, K, stream))
res["two_kernels_us"] = t(two); res["fused_us"] = t(one); res["two_kernels_us_again"] = t(two); res["fused_us_again"] = t(one)
res["tune_14"] = os.environ.get("DIM_TUNE14", "32")
print(json.dumps(res))
