"""GPU: the streaming K loop prototype of the small-problem GEMM (gemm_x6_stream_kernel, round 5; research library, dim_tune_set(14, 63)) against the
staged loop of the product at the shapes of ONE LightGlue pair (4096 rows = 2 x 2048 keypoints; q|k|v 256 -> 768, out-projection 256 -> 256 + residual,
ffn.0 512 -> 512, ffn.3 512 -> 256 + residual): bit equality, error vs fp64, time per launch.  Result: profiles/r05_ab_small_gemm_stream.jsonl (slower).
DIM_LIB=<research variant built with another -DDIM_STREAM_D> runs the prototype of that library only."""
import ctypes, importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi')
build = importlib.import_module('deep-image-matching_amd.build')
variant = os.environ.get("DIM_LIB")
lib = capi.load(variant if variant else str(build.LIBDIR / "libdim_hip_research.so"))
capi.install(lib, None)
p = lambda t: ctypes.c_void_p(t.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(5)


def t_us(fn, reps=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)


res = {"library": os.path.basename(variant) if variant else "libdim_hip_research.so", "shapes": []}
M = 4096
for K, N, with_r in ((256, 768, False), (256, 256, True), (512, 512, False), (512, 256, True), (256, 512, False)):
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(K, N, generator=g) / K ** 0.5).contiguous()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if with_r else None
    h, npad = ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W), K, N, ctypes.byref(h), ctypes.byref(npad)) == 0
    outs, times = {}, {}
    for name, kc in (("staged", 0), ("stream", 63)):
        if variant and name == "staged":
            continue
        assert lib.dim_tune_set(14, kc) == 0, lib.dim_last_error()
        C = torch.full((M, N), -3.0).cuda()
        run = lambda: capi.check(lib, lib.dim_op_gemm_x6_f32(p(A), K, h, npad.value, p(bias), p(R) if with_r else None, N, p(C), N, M, N, K, 0, stream))
        run(); torch.cuda.synchronize()
        outs[name] = C.clone()
        times[name] = t_us(run)
    lib.dim_tune_set(14, 0)
    ref = A.double().cpu() @ W.double() + bias.double().cpu() + (R.double().cpu() if with_r else 0)
    mag = A.abs().double().cpu() @ W.abs().double()
    rec = {"K": K, "N": N, "residual": with_r, "us": times, "rel_err_vs_fp64": float(((outs["stream"].double().cpu() - ref).abs() / mag).max())}
    if "staged" in outs:
        rec["bit_equal_staged_vs_stream"] = bool(torch.equal(outs["staged"], outs["stream"]))
    res["shapes"].append(rec)
    lib.dim_x3_destroy(h)

print(json.dumps(res))
