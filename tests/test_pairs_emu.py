"""CPU: pair generation (pairs_generator.py restated; matching_lowres on the device) through the emulator."""
import importlib

import numpy as np
import torch

from oracle import lightglue_ref, superpoint_ref, tile_ref

pairs_mod = importlib.import_module("deep-image-matching_amd.pairs")
weights = importlib.import_module("deep-image-matching_amd.weights")


def test_sequential_and_bruteforce_follow_the_reference():
    names = ["a", "b", "c", "d"]
    assert pairs_mod.pairs_from_sequential(names, 2) == [("a", "b"), ("a", "c"), ("b", "c"), ("b", "d"), ("c", "d")]
    assert pairs_mod.pairs_from_sequential(names, 0) == []
    assert pairs_mod.pairs_from_bruteforce(names) == [("a", "b"), ("a", "c"), ("a", "d"), ("b", "c"), ("b", "d"), ("c", "d")]


def test_lowres_pair_selection_matches_the_oracle_pipeline(emu_lib):
    rng = np.random.default_rng(5)
    base = (rng.random((90, 120)) * 255).astype(np.float32)
    # the third image is SMALLER than resize_max: the reference up-samples it (cv2.resize INTER_AREA, bilinear emulation)
    images = [base, np.roll(base, 7, axis=1).copy(), (rng.random((40, 52)) * 255).astype(np.float32), base[::-1].copy()]
    names = ["i0.jpg", "i1.jpg", "i2.jpg", "i3.jpg"]
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(0), weights.synthetic_lightglue_state_dict(0, 256)
    old_sp, old_lg = dict(pairs_mod.LOWRES_SP_CONF), dict(pairs_mod.LOWRES_LG_CONF)
    pairs_mod.LOWRES_SP_CONF.update(max_keypoints=48, nms_radius=2)  # emulator-sized
    pairs_mod.LOWRES_LG_CONF.update(n_layers=2, filter_threshold=0.0)
    try:
        sel = pairs_mod.LowresPairSelector(sp_sd, lg_sd, resize_max=64, min_matches=3, pair_batch=4, device="cpu", lib=emu_lib)
        table = sel.extract(images)
        idx_pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
        counts = sel.match_counts(table, idx_pairs)

        feats = []
        for im in images:
            _, _, new = tile_ref.preselection_sizes(im.shape, 64)
            small = tile_ref.resize_area(im, new) / np.float32(255.0)
            feats.append(superpoint_ref.superpoint_forward(torch.from_numpy(small)[None, None], sp_sd, sel._sp.cfg))
        ref = []
        for i, j in idx_pairs:
            k0, k1 = feats[i]["keypoints"].float(), feats[j]["keypoints"].float()
            s0, s1 = 1 + k0.max(0).values - k0.min(0).values, 1 + k1.max(0).values - k1.min(0).values
            r = lightglue_ref.lightglue_forward(k0, feats[i]["descriptors"].t().contiguous(), s0, k1, feats[j]["descriptors"].t().contiguous(), s1,
                                                lg_sd, dict(pairs_mod.LOWRES_LG_CONF))
            ref.append(len(r["matches"]))
        assert counts.tolist() == ref and max(ref) > 3
        got = sel.select(names, images)
        assert got == [(names[i], names[j]) for (i, j), c in zip(idx_pairs, ref) if c > 3]
        assert sel.select(names[:1], images[:1]) == []
    finally:
        pairs_mod.LOWRES_SP_CONF.clear(); pairs_mod.LOWRES_SP_CONF.update(old_sp)
        pairs_mod.LOWRES_LG_CONF.clear(); pairs_mod.LOWRES_LG_CONF.update(old_lg)


def _reference_pairs_from_score_matrix():
    """The reference's own function, executed from its source (hloc/pairs_from_retrieval.py imports h5py at module level)."""
    import ast
    from pathlib import Path

    src = Path("/root/reference/src/deep_image_matching/thirdparty/hloc/pairs_from_retrieval.py")
    if not src.exists():
        return None
    fn = [n for n in ast.parse(src.read_text()).body if isinstance(n, ast.FunctionDef) and n.name == "pairs_from_score_matrix"][0]
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "pairs_from_retrieval", "exec"), ns)
    return ns["pairs_from_score_matrix"]


def test_retrieval_pairs_match_the_reference_function(emu_lib):
    """einsum + masking + top-k on the device vs thirdparty/hloc/pairs_from_retrieval.py:49-70 (the reference's own function
    when /root/reference exists, its restatement otherwise)."""
    g = torch.Generator().manual_seed(3)
    names = [f"img{i:02d}.jpg" for i in range(37)]
    desc = torch.nn.functional.normalize(torch.randn(37, 96, generator=g), dim=-1)
    desc[5] = -desc[9]                      # a strongly negative pair: removed by min_score = 0
    got = pairs_mod.pairs_from_retrieval(names, names, desc.numpy(), desc.numpy(), num_matched=6, device="cpu", lib=emu_lib)
    sim = torch.einsum("id,jd->ij", desc, desc)
    self_mask = np.array(names)[:, None] == np.array(names)[None]
    ref_fn = _reference_pairs_from_score_matrix()
    if ref_fn is not None:
        ref = [(names[i], names[j]) for i, j in ref_fn(sim.clone(), self_mask.copy(), 6, min_score=0)]
    else:
        s = sim.clone().masked_fill_(torch.from_numpy(self_mask) | (sim < 0), float("-inf"))
        tk = torch.topk(s, 6, dim=1)
        ref = [(names[i], names[int(tk.indices[i, j])]) for i in range(37) for j in range(6) if torch.isfinite(tk.values[i, j])]
    assert got == ref and len(got) > 100 and all(a != b for a, b in got) and ("img05.jpg", "img09.jpg") not in got
    # fewer valid candidates than num_matched: only the finite ones are emitted
    few = pairs_mod.pairs_from_retrieval(names[:3], names[:3], desc[:3].numpy(), desc[:3].numpy(), num_matched=5, min_score=None, device="cpu", lib=emu_lib)
    assert len(few) == 6 and all(a != b for a, b in few)
