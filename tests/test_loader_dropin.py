"""CPU (build container only): the LOADER-level drop-in (SURVEY §8b "discovery").

The three ``*_mi355x.py`` modules of INTEGRATION.md §2 are written — verbatim as the document shows them — into a temporary
directory that is appended to the ``__path__`` of the (stub) ``deep_image_matching.extractors`` / ``.matchers`` packages, the
zoo entry of INTEGRATION.md §2 is added to the reference's own ``config.confs`` / ``opt_zoo``, and then the REFERENCE's code does
the rest, unmodified, from /root/reference:

    Config(args)                                              config.py
    ImageMatcher(config)      -> extractor_loader / matcher_loader  (extractor_base.py:29-52, matcher_base.py:36-60:
                                 exactly one subclass DEFINED IN the module) -> plugin constructors     image_matching.py:280-321
    ImageMatcher.generate_pairs() -> PairsGenerator bruteforce                                          pairs_generator.py
    ImageMatcher.extract_features() / .match_pairs()           the two hot loops                        image_matching.py:413-494

with the emulator build of the kernels.  The artefacts must equal what BatchedImageMatcher writes for the same images."""
import importlib
import re
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

from tests import refstubs
from tests.test_reference_base_classes import _write_images

pytestmark = pytest.mark.skipif(not refstubs.available(), reason="/root/reference not present")
PKG = "deep_image_matching"
ROOT = Path(__file__).resolve().parents[1]


def _integration_md_modules():
    """{relative path: source} of the python blocks INTEGRATION.md §2 introduces with a `src/deep_image_matching/...py` line."""
    text = (ROOT / "INTEGRATION.md").read_text()
    out = {}
    for m in re.finditer(r"`src/deep_image_matching/((?:extractors|matchers)/\w+_mi355x\.py)`.*?```python\n(.*?)```", text, re.S):
        out[m.group(1)] = m.group(2)
    return out


@pytest.fixture
def dim_tree(emu_install, tmp_path):
    added = refstubs.install(find_fundamental=lambda p0, p1, *a: (np.eye(3), np.ones((len(p0), 1), np.uint8)))
    for k in [k for k in sys.modules if k.startswith("deep-image-matching_amd.plugins")]:
        del sys.modules[k]
    try:
        plugins = importlib.import_module("deep-image-matching_amd.plugins")
        assert plugins.HAVE_DIM
        # what the real package __init__ files export and image_matching.py imports (image_matching.py:26-38); the heavyweight
        # __init__ files themselves (every other extractor / matcher) are not executed
        eb = importlib.import_module(PKG + ".extractors.extractor_base")
        mb = importlib.import_module(PKG + ".matchers.matcher_base")
        ex, ma, io, ut = (sys.modules[PKG + s] for s in (".extractors", ".matchers", ".io", ".utils"))
        ex.extractor_loader, ex.SuperPointExtractor = eb.extractor_loader, plugins.SuperPointExtractor
        ma.matcher_loader, ma.LightGlueMatcher = mb.matcher_loader, plugins.LightGlueMatcher
        io.get_features = importlib.import_module(PKG + ".io.h5").get_features
        kf = types.ModuleType("kornia.feature")
        sys.modules["kornia.feature"] = kf
        sys.modules["kornia"].feature = kf
        added.append("kornia.feature")
        ut.ImageList = importlib.import_module(PKG + ".utils.image").ImageList
        ut.get_pairs_from_file = importlib.import_module(PKG + ".utils.utils").get_pairs_from_file
        # the drop-in: INTEGRATION.md's module files, on the packages' search path
        mods = _integration_md_modules()
        assert set(mods) == {"extractors/superpoint_mi355x.py", "extractors/aliked_mi355x.py", "matchers/lightglue_mi355x.py"}, sorted(mods)
        for rel, src in mods.items():
            p = tmp_path / "dropin" / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(src)
        ex.__path__.append(str(tmp_path / "dropin" / "extractors"))
        ma.__path__.append(str(tmp_path / "dropin" / "matchers"))
        config = importlib.import_module(PKG + ".config")
        yield plugins, config, tmp_path
    finally:
        sys.modules.pop("kornia.feature", None)
        refstubs.uninstall(added)
        for k in [k for k in sys.modules if k.startswith("deep-image-matching_amd.plugins")]:
            del sys.modules[k]
        importlib.import_module("deep-image-matching_amd.plugins")


def _zoo_entry(config, project):
    """INTEGRATION.md §2's zoo entry (test-sized parameters; checkpoints on disk in the official key layout)."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    torch.save(weights.synthetic_superpoint_state_dict(1234), project / "sp.pth")
    torch.save(weights.synthetic_lightglue_state_dict(0, 256, n_layers=2, gain=1.0), project / "lg.pth")
    config.confs["superpoint_mi355x+lightglue_mi355x"] = {
        "extractor": {"name": "superpoint_mi355x", "nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 300, "remove_borders": 2,
                      "weights_path": str(project / "sp.pth")},
        "matcher": {"name": "lightglue_mi355x", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0,
                    "weights_path": str(project / "lg.pth"), "pruning_min_kpts": -1},
    }
    if "superpoint_mi355x" not in config.opt_zoo["extractors"]:
        config.opt_zoo["extractors"].append("superpoint_mi355x")
        config.opt_zoo["matchers"].append("lightglue_mi355x")


def test_loaders_discover_the_dropin_modules_and_the_reference_loops_run(dim_tree):
    plugins, config, project = dim_tree
    imgs = _write_images(project / "images", n=3)
    _zoo_entry(config, project)
    import yaml
    yml = project / "user.yaml"
    yml.write_text(yaml.safe_dump({"general": {"geom_verification": "NONE", "min_inliers_per_pair": 1, "min_inlier_ratio_per_pair": 0.0}}))
    cfg = config.Config({"dir": str(project), "pipeline": "superpoint_mi355x+lightglue_mi355x", "strategy": "bruteforce", "tiling": "none",
                         "force": True, "config_file": str(yml), "outs": str(project / "out_loader")})
    # discovery exactly as ImageMatcher.__init__ does it (image_matching.py:303-321)
    eb = importlib.import_module(PKG + ".extractors.extractor_base")
    mb = importlib.import_module(PKG + ".matchers.matcher_base")
    E = eb.extractor_loader(sys.modules[PKG + ".extractors"], cfg.extractor["name"])
    M = mb.matcher_loader(sys.modules[PKG + ".matchers"], cfg.matcher["name"])
    A = eb.extractor_loader(sys.modules[PKG + ".extractors"], "aliked_mi355x")
    assert E.__name__ == "SuperPointMI355XExtractor" and E.__module__ == PKG + ".extractors.superpoint_mi355x" and issubclass(E, plugins.SuperPointExtractor)
    assert M.__name__ == "LightGlueMI355XMatcher" and issubclass(M, plugins.LightGlueMatcher) and issubclass(M, mb.MatcherBase)
    assert A.__name__ == "AlikedMI355XExtractor" and issubclass(A, eb.ExtractorBase) and A.descriptor_size == 128 and A.grayscale is False
    # the reference's ImageMatcher end to end
    imm = importlib.import_module(PKG + ".image_matching")
    matcher = imm.ImageMatcher(cfg)
    assert type(matcher._extractor) is E and type(matcher._matcher) is M
    matcher.generate_pairs()
    assert len(matcher.pairs) == 3
    feature_path = matcher.extract_features()
    matches_path = matcher.match_pairs(feature_path)
    h5 = importlib.import_module(PKG + ".io.h5")
    out = cfg.general["output_dir"]
    assert feature_path == out / "features.h5" and matches_path == out / "matches.h5" and (out / "raw_matches.h5").exists()
    # the batched loops on the SAME plugin instances write the same artefacts (INTEGRATION.md §2 "Batched loops")
    bm_mod = importlib.import_module("deep-image-matching_amd.batched_matcher")
    export = importlib.import_module("deep-image-matching_amd.export")
    importlib.reload(export)   # h5py look-alike is installed: the batched writers use the .h5 containers too
    try:
        bm_mod.export = export
        bm = bm_mod.BatchedImageMatcher(matcher._extractor, matcher._matcher, project / "out_batched", image_batch=2, pair_batch=2, verify=False)
        fp2 = bm.extract_features(imgs)
        mp2 = bm.match_pairs(fp2, [(a.name, b.name) for a, b in ((imgs[0], imgs[1]), (imgs[0], imgs[2]), (imgs[1], imgs[2]))])
        n_matches = 0
        for p in imgs:
            a, b = h5.get_features(feature_path, p.name), h5.get_features(fp2, p.name)
            assert a["keypoints"].shape == (300, 2) and set(a) == set(b)
            for k in a:
                assert np.array_equal(a[k], b[k]), (p.name, k)
        for a, b in ((imgs[0], imgs[1]), (imgs[0], imgs[2]), (imgs[1], imgs[2])):
            r1, r2 = h5.get_matches(out / "raw_matches.h5", a.name, b.name), h5.get_matches(project / "out_batched" / "raw_matches.h5", a.name, b.name)
            v1, v2 = h5.get_matches(matches_path, a.name, b.name), h5.get_matches(mp2, a.name, b.name)
            assert np.array_equal(np.asarray(r1), np.asarray(r2)) and np.array_equal(np.asarray(v1), np.asarray(v2)) and np.asarray(r1).dtype == np.int64
            n_matches += len(np.asarray(v1))
        assert n_matches >= 24
    finally:
        refstubs_h5 = sys.modules.pop("h5py", None)
        importlib.reload(export)
        bm_mod.export = export
        if refstubs_h5 is not None:
            sys.modules["h5py"] = refstubs_h5
