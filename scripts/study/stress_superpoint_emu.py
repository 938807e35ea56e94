"""CPU sweep (no GPU): random SuperPoint configurations — image sizes that are not multiples of 8, NMS radii 1-5, thresholds, top-k / keep-all,
border widths, both descriptor samplers, plateau images — through the HIP sources on the test emulator against the oracle: NMS bit-exact on our score
map, selection equal to the oracle's on our NMS map, end result through compare_superpoint.   python scripts/study/stress_superpoint_emu.py SEED N"""
import importlib, sys, random, torch, ctypes
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import superpoint_ref
from tests.parity import compare_superpoint
build = importlib.import_module("deep-image-matching_amd.build")
sp_mod = importlib.import_module("deep-image-matching_amd.superpoint_hip")
weights = importlib.import_module("deep-image-matching_amd.weights")
lib = ctypes.CDLL(str(build.build_emu())); lib.dim_last_error.restype = ctypes.c_char_p
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for it in range(N):
    H, W = rnd.randint(16, 90), rnd.randint(16, 110)
    cfg = {"nms_radius": rnd.choice([1, 2, 3, 4, 5]), "keypoint_threshold": rnd.choice([0.0005, 0.002, 0.005]), "max_keypoints": rnd.choice([-1, 5, 40, 200]),
           "remove_borders": rnd.choice([0, 2, 4]), "fix_sampling": rnd.choice([False, True])}
    sd = weights.synthetic_superpoint_state_dict(rnd.randrange(50))
    img = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(rnd.randrange(10000)))
    if rnd.random() < 0.2: img = (img * 8).floor() / 8      # plateaus
    try:
        net = sp_mod.SuperPointHIP(sd, cfg, max_batch=1, max_hw=(H, W), capacity=(H * W if cfg["max_keypoints"] < 0 else 1024), device="cpu", lib=lib)
        out = {k: v.cpu() for k, v in net(img).items()}
        taps = net.debug_taps()
        ref = superpoint_ref.superpoint_forward(img, sd, cfg, taps=True)
        nms_on_ours = superpoint_ref.simple_nms(taps["score_map"], cfg["nms_radius"])
        assert torch.equal(nms_on_ours[0], taps["nms_map"][0]), "nms"
        yx, sc = superpoint_ref.select_keypoints(taps["nms_map"][0], cfg["keypoint_threshold"], cfg["remove_borders"], cfg["max_keypoints"])
        assert set(map(tuple, torch.flip(yx, [1]).tolist())) == set(map(tuple, out["keypoints"].long().tolist())), "selection on our map"
        r = {"keypoints": ref["keypoints"], "scores": ref["scores"], "descriptors": ref["descriptors"]}
        compare_superpoint(out, r)
    except Exception as e:
        bad += 1; print("FAIL", it, dict(H=H, W=W, cfg=cfg), repr(e)[:400], flush=True)
print("done", N, "failures", bad)
