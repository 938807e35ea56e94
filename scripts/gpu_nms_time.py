"""GPU: simple_nms per 512x512 score map (the 1024^2 bench image), 32x32 vs 64x64 tiles (dim_tune_set key 7), bit-identical."""
import ctypes, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); p = capi.ptr
B = 100
x = torch.rand(B, 512, 512, device='cuda'); out = torch.empty_like(x)
def run(r): capi.check(lib, lib.dim_op_simple_nms_f32(p(x), p(out), B, 512, 512, r, None))
for r in (1, 2, 3, 4):
    res, ref = {}, None
    for big in (0, 1, 0, 1):
        lib.dim_tune_set(7, big)
        run(r); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(r)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(big, []).append(round(e0.elapsed_time(e1) / 10 / B * 1000, 2))
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out)
    print('nms r', r, 'us per 512^2 map: 32-tiles', res[0], '64-tiles', res[1])
lib.dim_tune_set(7, 1)
