"""GPU (round 6): the small-problem GEMM block of ONE LightGlue pair (4096 rows) — the product's 64 x 128 block with waves 2 x 2 against the
research variants selected by dim_tune_set(14, .): 71 = 64 x 128 with waves 1 x 4 (pipelined K loop, every weight fragment fetched by one wave),
72 = 32 x 128 with waves 1 x 4 (twice the workgroups), 74 = 64 x 256 with waves 1 x 4.  Bit equality with the product block, us per launch."""
import ctypes, importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi')
build = importlib.import_module('deep-image-matching_amd.build')
lib = capi.load(str(build.LIBDIR / "libdim_hip_research.so"))
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


res = {"shapes": []}
M = 4096
for K, N, with_r in ((256, 768, False), (256, 512, False), (512, 512, False), (512, 256, True), (256, 256, False)):
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(K, N, generator=g) / K ** 0.5).contiguous()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if with_r else None
    h, npad = ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W), K, N, ctypes.byref(h), ctypes.byref(npad)) == 0
    outs, times = {}, {}
    for name, kc in (("product_32x128_chunk_ahead", 0), ("32x128_kc64_step_pipelined", 79), ("64x128_1x4", 71), ("product_round5_64x128_2x2", 70), ("32x128_kc32", 78), ("32x128_dbuf", 76), ("64x256_1x4", 74), ("product_again", 0)):
        assert lib.dim_tune_set(14, kc) == 0, lib.dim_last_error()
        C = torch.full((M, N), -3.0).cuda()
        run = lambda: capi.check(lib, lib.dim_op_gemm_x6_f32(p(A), K, h, npad.value, p(bias), p(R) if with_r else None, N, p(C), N, M, N, K, 0, stream))
        run(); torch.cuda.synchronize()
        outs[name] = C.clone()
        times[name] = t_us(run)
    lib.dim_tune_set(14, 0)
    rec = {"K": K, "N": N, "residual": with_r, "us": times,
           "bit_equal_to_product": {k: bool(torch.equal(outs["product_32x128_chunk_ahead"], v)) for k, v in outs.items() if k != "product_32x128_chunk_ahead"}}
    res["shapes"].append(rec)
    lib.dim_x3_destroy(h)
print(json.dumps(res))
