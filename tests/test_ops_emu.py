"""CPU: operator-level kernels on the test emulator vs torch / the oracle (bit-exact where integer)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import superpoint_ref


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("radius", [0, 1, 2, 3, 4, 5, 6])
def test_simple_nms_bit_exact_with_ties_and_plateaus(emu_lib, radius):
    g = torch.Generator().manual_seed(radius)
    H, W = 45, 70  # not multiples of the tile; includes the image border logic
    s = torch.rand(2, H, W, generator=g)
    s[0] = (s[0] * 6).round() / 6 + 0.01  # heavy ties / plateaus (SURVEY App. D KATs)
    s[1, 10:20, 10:30] = 0.5
    out = torch.full_like(s, -1.0)
    assert emu_lib.dim_op_simple_nms_f32(p(s), p(out), 2, H, W, radius, None) == 0, emu_lib.dim_last_error()
    ref = superpoint_ref.simple_nms(s, radius)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("radius", [1, 3, 4])
def test_simple_nms_64_tiles_bit_exact(emu_lib, radius):
    """The 64 x 64-tile variant (dim_tune_set key 7 = 2 forces it on any map size) on a map with partial tiles on both axes."""
    g = torch.Generator().manual_seed(40 + radius)
    H, W = 75, 130
    s = torch.rand(1, H, W, generator=g)
    s[0, :40] = (s[0, :40] * 5).round() / 5 + 0.01
    s[0, 50:70, 60:100] = 0.5
    out = torch.full_like(s, -1.0)
    try:
        emu_lib.dim_tune_set(7, 2)
        assert emu_lib.dim_op_simple_nms_f32(p(s), p(out), 1, H, W, radius, None) == 0, emu_lib.dim_last_error()
    finally:
        emu_lib.dim_tune_set(7, 1)
    assert torch.equal(out, superpoint_ref.simple_nms(s, radius))


@pytest.mark.parametrize("M,N,K,bt", [(200, 65, 64, 0), (130, 256, 256, 0), (150, 140, 64, 1), (1, 4, 32, 0)])
def test_gemm_mfma(emu_lib, M, N, K, bt):
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if bt else torch.randn(K, ((N + 3) // 4) * 4, generator=g)
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    C = torch.zeros(M, N)
    assert emu_lib.dim_op_gemm_f32(p(A), K, p(B), B.shape[1], bt, p(bias), p(R), N, p(C), N, M, N, K, 1, None) == 0
    ref = torch.relu((A @ B.T if bt else A @ B[:, :N]) + bias + R)
    assert (C - ref).abs().max() < 1e-4


@pytest.mark.parametrize("cin,cout,H,W,pool", [(64, 64, 20, 37, 1), (64, 128, 9, 33, 0), (128, 128, 16, 34, 1)])
def test_conv3x3_mfma(emu_lib, cin, cout, H, W, pool):
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    xin = x.permute(0, 2, 3, 1).contiguous()
    wk = w.permute(2, 3, 1, 0).contiguous().reshape(9, cin, cout)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.full((2, Ho, Wo, cout), -7.0)
    assert emu_lib.dim_op_conv3x3_nhwc_f32(p(xin), p(wk), p(b), p(out), 2, H, W, cin, cout, pool, 1, None) == 0
    ref = torch.relu(F.conv2d(x, w, b, padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    assert (out - ref.permute(0, 2, 3, 1)).abs().max() < 1e-4


@pytest.fixture(params=[2, 1], ids=["fp16x3", "bf16x6"])
def split_mode(request, emu_lib):
    """Both split-precision modes of the matrix-core kernels (dim_tune_set key 1); the default is restored."""
    emu_lib.dim_tune_set(1, request.param)
    yield request.param
    emu_lib.dim_tune_set(1, 2)


@pytest.mark.parametrize("M,N,K", [(200, 65, 64), (130, 256, 512), (64, 130, 32)])
def test_gemm_split_precision_is_fp32_accurate(emu_lib, split_mode, M, N, K):
    """bf16x6 (exact 3-way bf16 split, six cross terms) and fp16x3 (2-way fp16 split of the scaled operands,
    three cross terms) on the 16-bit MFMA == fp32-class accuracy (vs fp64)."""
    g = torch.Generator().manual_seed(K + M)
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g).contiguous()
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    C = torch.zeros(M, N)
    dev, npad = ctypes.c_void_p(), ctypes.c_int()
    assert emu_lib.dim_x3_create(p(W), K, N, ctypes.byref(dev), ctypes.byref(npad)) == 0
    assert emu_lib.dim_op_gemm_x6_f32(p(A), K, dev, npad.value, p(bias), p(R), N, p(C), N, M, N, K, 0, None) == 0, emu_lib.dim_last_error()
    emu_lib.dim_x3_destroy(dev)
    ref = (A.double() @ W.double() + bias.double() + R.double())
    mag = (A.abs().double() @ W.abs().double())
    assert ((C.double() - ref).abs() / mag).max().item() < 4e-7


@pytest.mark.parametrize("M,N,K", [(130, 256, 512), (200, 500, 64)])
def test_gemm_wide_blocks_are_bit_identical(emu_lib, M, N, K):
    """The 128 x 256 workgroup block of gemm_x6.hip (dim_tune_set key 6; 2 = forced whatever the problem size) accumulates
    every output in the same order as the 128 x 128 block."""
    g = torch.Generator().manual_seed(K + M)
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g).contiguous()
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    dev, npad = ctypes.c_void_p(), ctypes.c_int()
    assert emu_lib.dim_x3_create(p(W), K, N, ctypes.byref(dev), ctypes.byref(npad)) == 0
    outs = []
    try:
        for wide in (0, 2):
            emu_lib.dim_tune_set(6, wide)
            C = torch.full((M, N), -3.0)
            assert emu_lib.dim_op_gemm_x6_f32(p(A), K, dev, npad.value, p(bias), p(R), N, p(C), N, M, N, K, 1, None) == 0, emu_lib.dim_last_error()
            outs.append(C)
    finally:
        emu_lib.dim_tune_set(6, 1)
        emu_lib.dim_x3_destroy(dev)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(130, 256, 512), (200, 500, 64), (70, 128, 256)])
def test_gemm_streaming_k_loop_prototype_is_bit_identical(emu_research_lib, M, N, K):
    """Research build only (dim_tune_set key 14 = 63): the small-problem block with the barrier-free streaming K loop (activation fragments straight from
    global memory, register ring) — same pieces, same term order as the staged loop; measured slower on hardware (profiles/r05_ab_small_gemm_stream.jsonl)."""
    lib = emu_research_lib
    g = torch.Generator().manual_seed(K + M)
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g).contiguous()
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    dev, npad = ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W), K, N, ctypes.byref(dev), ctypes.byref(npad)) == 0
    outs = []
    try:
        for kc in (0, 63):
            assert lib.dim_tune_set(14, kc) == 0
            C = torch.full((M, N), -3.0)
            assert lib.dim_op_gemm_x6_f32(p(A), K, dev, npad.value, p(bias), p(R), N, p(C), N, M, N, K, 1, None) == 0, lib.dim_last_error()
            outs.append(C)
    finally:
        lib.dim_tune_set(14, 0)
        lib.dim_x3_destroy(dev)
    assert torch.equal(outs[0], outs[1]) and (outs[0] != -3.0).all()


@pytest.mark.parametrize("cin,cout,H,W,pool", [(64, 64, 20, 37, 1), (64, 128, 9, 33, 0), (128, 128, 16, 34, 1)])
def test_conv3x3_split_precision_is_fp32_accurate(emu_lib, split_mode, cin, cout, H, W, pool):
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=g)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).contiguous()
    b = torch.randn(cout, generator=g)
    xin = x.permute(0, 2, 3, 1).contiguous()
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.full((2, Ho, Wo, cout), -7.0)
    dev = ctypes.c_void_p()
    assert emu_lib.dim_convx6_create(p(w), cin, cout, ctypes.byref(dev)) == 0
    assert emu_lib.dim_op_conv3x3_x6_nhwc_f32(p(xin), dev, p(b), p(out), 2, H, W, cin, cout, pool, 1, None) == 0
    emu_lib.dim_x3_destroy(dev)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    mag = F.conv2d(x.abs().double(), w.abs().double(), padding=1).max().item()
    assert (out.double() - ref.permute(0, 2, 3, 1)).abs().max().item() / mag < 4e-7


@pytest.mark.parametrize("regime", ["tiny", "large", "overflow"])
def test_fp16x3_range_behaviour(emu_lib, regime):
    """fp16x3's narrow exponent: activations are scaled by 16 and clamped to +-65504 before the split.
    tiny (1e-3): still ~1e-6-accurate relative to sum|a||b|; large (up to 4000): exact range; beyond 4094:
    saturates to a finite value (never inf/nan)."""
    emu_lib.dim_tune_set(1, 2)
    g = torch.Generator().manual_seed(3)
    M, N, K = 64, 64, 256
    scale = {"tiny": 1e-3, "large": 4000.0, "overflow": 1e6}[regime]
    A = (torch.rand(M, K, generator=g) * scale).contiguous()
    W = (torch.randn(K, N, generator=g) * 0.05).contiguous()
    C = torch.zeros(M, N)
    dev, npad = ctypes.c_void_p(), ctypes.c_int()
    assert emu_lib.dim_x3_create(p(W), K, N, ctypes.byref(dev), ctypes.byref(npad)) == 0
    assert emu_lib.dim_op_gemm_x6_f32(p(A), K, dev, npad.value, None, None, 0, p(C), N, M, N, K, 0, None) == 0
    emu_lib.dim_x3_destroy(dev)
    assert torch.isfinite(C).all()
    if regime != "overflow":
        ref, mag = A.double() @ W.double(), A.abs().double() @ W.abs().double()
        assert ((C.double() - ref).abs() / mag).max().item() < (2e-6 if regime == "tiny" else 4e-7)


def _ffn_ln_gelu_case(lib, M, K, seed, device="cpu"):
    """gelu(layer_norm(A W + b)) through dim_op_gemm_x6_ln_gelu_f32 -> (device result, fp64 reference)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g) * 1.5
    W = (torch.randn(K, 512, generator=g) / K ** 0.5).contiguous()
    bias, gamma, beta = torch.randn(512, generator=g) * 0.1, 1.0 + 0.2 * torch.randn(512, generator=g), 0.1 * torch.randn(512, generator=g)
    dev, npad = ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W), K, 512, ctypes.byref(dev), ctypes.byref(npad)) == 0 and npad.value == 512
    Ad, bd, gd, btd = (t.to(device).contiguous() for t in (A, bias, gamma, beta))
    C = torch.full((M, 512), -7.0, device=device)
    try:
        rc = lib.dim_op_gemm_x6_ln_gelu_f32(p(Ad), K, dev, p(bd), p(gd), p(btd), p(C), 512, M, K, None)
        assert rc == 0, lib.dim_last_error()
        if device != "cpu":
            torch.cuda.synchronize()
    finally:
        lib.dim_x3_destroy(dev)
    h = A.double() @ W.double() + bias.double()
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(h, (512,), gamma.double(), beta.double(), 1e-5))
    return C.cpu(), ref


@pytest.mark.parametrize("M,K", [(64, 512), (150, 512), (67, 256)])
def test_ffn_layernorm_gelu_fused_op_vs_fp64(emu_lib, M, K):
    """LightGlue's ffn.0 -> LayerNorm -> GELU as one kernel (64 x 512 blocks; ragged last block) against an fp64 evaluation."""
    C, ref = _ffn_ln_gelu_case(emu_lib, M, K, seed=M + K)
    assert (C.double() - ref).abs().max().item() < 5e-6


def _ffn_fused_case(lib, M, K, seed, device="cpu"):
    """residual + gelu(layer_norm(A W0 + b0)) W3 + b3 through dim_op_ffn_fused_f32 -> (device result, fp64 reference)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g) * 1.5
    W0 = (torch.randn(K, 512, generator=g) / K ** 0.5).contiguous()
    W3 = (torch.randn(512, 256, generator=g) / 512 ** 0.5).contiguous()
    b0, gamma, beta = torch.randn(512, generator=g) * 0.1, 1.0 + 0.2 * torch.randn(512, generator=g), 0.1 * torch.randn(512, generator=g)
    b3, R = torch.randn(256, generator=g) * 0.1, torch.randn(M, 256, generator=g)
    h0, h3, npad = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
    assert lib.dim_x3_create(p(W0), K, 512, ctypes.byref(h0), ctypes.byref(npad)) == 0 and npad.value == 512
    assert lib.dim_x3_create_kperm(p(W3), 512, 256, ctypes.byref(h3), ctypes.byref(npad)) == 0 and npad.value == 256
    Ad, b0d, gd, btd, b3d, Rd = (t.to(device).contiguous() for t in (A, b0, gamma, beta, b3, R))
    C = torch.full((M, 256), -7.0, device=device)
    try:
        rc = lib.dim_op_ffn_fused_f32(p(Ad), K, h0, p(b0d), p(gd), p(btd), h3, p(b3d), p(Rd), 256, p(C), 256, M, K, None)
        assert rc == 0, lib.dim_last_error()
        if device != "cpu":
            torch.cuda.synchronize()
    finally:
        lib.dim_x3_destroy(h0); lib.dim_x3_destroy(h3)
    h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(A.double() @ W0.double() + b0.double(), (512,), gamma.double(), beta.double(), 1e-5))
    return C.cpu(), R.double() + h @ W3.double() + b3.double()


@pytest.mark.parametrize("M,K", [(64, 512), (150, 512), (67, 256)])
def test_ffn_fused_op_vs_fp64(emu_lib, M, K):
    """ffn.0 -> LayerNorm -> GELU -> ffn.3 + residual as one kernel (hidden tile register-resident; ragged last block) vs fp64."""
    C, ref = _ffn_fused_case(emu_lib, M, K, seed=M + K)
    assert (C.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("M,N,K", [(150, 200, 256), (128, 128, 64), (37, 300, 256)])
def test_gemm_x6_nt_vs_fp64(emu_lib, M, N, K):
    """sim = A B^T with both operands split on the fly (gemm_x6_nt_kernel, LightGlue's similarity) vs fp64; ragged last blocks; a
    guard band around C stays untouched."""
    g = torch.Generator().manual_seed(M + N)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    ldc = N + 8
    C = torch.full((M + 3, ldc), -7.0)
    rc = emu_lib.dim_op_gemm_x6_nt_f32(p(A), K, p(B), K, p(C), ldc, M, N, K, None)
    assert rc == 0, emu_lib.dim_last_error()
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())
    assert ((C[:M, :N].double() - ref).abs() / scale).max().item() < 5e-7
    assert bool((C[M:] == -7.0).all()) and bool((C[:, N:] == -7.0).all())


@pytest.mark.parametrize("kf16,df16,dn", [(0, 0, 0), (1, 1, 1), (0, 1, 0), (1, 0, 1)])
def test_lg_stage_features_equals_the_host_conversion(emu_lib, kf16, df16, dn):
    """dim_lg_stage_features: one pair's arrays as features.h5 holds them (float16 or float32, descriptors (N, D) or (D, N)) -> the fp32 (N, D)
    feature table, bit for bit what featuresDict2Lightglue's host path produces (matchers/lightglue.py:38-43,62: transpose, torch.as_tensor(...,
    float32)); rows past the live count are zero; ragged counts incl. 0 and a count that is not a multiple of the 32-row block."""
    import importlib
    import numpy as np
    capi = importlib.import_module("deep-image-matching_amd.capi")
    g = np.random.default_rng(7 + kf16 + 2 * df16 + 4 * dn)
    D, cap = 128, 77
    for n0, n1 in ((77, 45), (0, 33), (64, 0)):
        arrs, descr, keep = [], [], []
        for n in (n0, n1):
            k = (g.random((n, 2)) * 1000).astype(np.float16 if kf16 else np.float32)
            d = g.standard_normal((D, n) if dn else (n, D)).astype(np.float16 if df16 else np.float32)
            kt, dt = torch.from_numpy(np.ascontiguousarray(k)), torch.from_numpy(np.ascontiguousarray(d))
            keep += [kt, dt]
            descr.append(capi.LgRawFeatures(kt.data_ptr(), dt.data_ptr(), n, kf16, df16, dn))
            arrs.append((k.astype(np.float32), (d.T if dn else d).astype(np.float32)))
        ktab, dtab = torch.full((2, cap, 2), -7.0), torch.full((2, cap, D), -7.0)
        assert emu_lib.dim_lg_stage_features(ctypes.byref(descr[0]), ctypes.byref(descr[1]), cap, D, p(ktab), p(dtab), None) == 0, emu_lib.dim_last_error()
        for i, n in enumerate((n0, n1)):
            assert np.array_equal(ktab[i, :n].numpy(), arrs[i][0]) and np.array_equal(dtab[i, :n].numpy(), arrs[i][1])
            assert not ktab[i, n:].any() and not dtab[i, n:].any()
