import ctypes, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); p = capi.ptr
x = torch.rand(16, 1024, 1024, device='cuda'); out = torch.empty_like(x)
def run(r): capi.check(lib, lib.dim_op_simple_nms_f32(p(x), p(out), 16, 1024, 1024, r, None))
for r in (3, 4, 5):
    run(r); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(r)
    e1.record(); torch.cuda.synchronize()
    print('nms r', r, 'us per image', round(e0.elapsed_time(e1) / 10 / 16 * 1000, 2))
