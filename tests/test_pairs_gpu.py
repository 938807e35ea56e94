"""GPU: matching_lowres pair generation at the reference's sizes (resize_max 1000, 2048 keypoints, 7 layers)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pairs_mod = importlib.import_module("deep-image-matching_amd.pairs")
weights = importlib.import_module("deep-image-matching_amd.weights")


def test_lowres_pairs_batched_equals_one_call_per_pair(hip_lib):
    rng = np.random.default_rng(0)
    base = (rng.random((1200, 1600)) * 255).astype(np.float32)
    images = [base, np.roll(base, 40, axis=1).copy(), (rng.random((1000, 1500)) * 255).astype(np.float32), base[:, ::-1].copy(),
              (rng.random((1600, 1200)) * 255).astype(np.float32)]
    names = [f"im{i}.jpg" for i in range(5)]
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(0), weights.synthetic_lightglue_state_dict(0, 256)
    a = pairs_mod.LowresPairSelector(sp_sd, lg_sd, pair_batch=8, lib=hip_lib)
    b = pairs_mod.LowresPairSelector(sp_sd, lg_sd, pair_batch=1, lib=hip_lib)
    ta = a.extract(images)
    assert ta[0].shape == (5, 2048, 2) and int(ta[2].min()) == 2048 and float(ta[3].max()) <= 1000.0
    idx = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    ca, cb = a.match_counts(ta, idx), b.match_counts(b.extract(images), idx)
    assert ca.shape == (10,) and np.array_equal(ca, cb)
    sel = a.select(names, images)
    assert sel == [(names[i], names[j]) for (i, j), c in zip(idx, ca) if c > 20]
