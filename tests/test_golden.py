"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py).
CPU: the oracle reproduces them on this machine (they were generated in the build container).
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from partsbaseddetector_amd import model as M, synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.astype(a.dtype).view(np.uint32))
    return np.array_equal(a, b)


def test_inputs_are_reproducible():
    assert np.int64(synth.synthetic_frame(1, 96, 80, 3).astype(np.int64).sum()) == G["tiny_frame_sum"]
    assert np.float64(M.synthetic_person_model().flatten().filters_f64.sum()) == G["person_filters_sum"]


def test_oracle_matches_golden(oracle):
    flat = M.synthetic_tiny_model(thresh=0.7).flatten()
    im = synth.synthetic_frame(1, 96, 80, 3)
    imgs, scales = oracle.pyramid_images(im, flat.sbin, flat.interval)
    feats, _ = oracle.features_pyramid(flat, im)
    assert _eq(scales, G["tiny_scales"]) and _eq(imgs[1], G["tiny_img1"]) and _eq(imgs[-1], G["tiny_img_last"])
    assert _eq(feats[0], G["tiny_feat0"]) and _eq(feats[-1], G["tiny_feat_last"])
    resp0 = oracle.responses(flat, feats[0])
    assert _eq(resp0, G["tiny_resp0"])
    Ix, Iy, Ik, rootv, rooti = oracle.dp_min(flat, 0, resp0)
    assert _eq(rootv, G["tiny_rootv0"]) and _eq(rooti, G["tiny_rooti0"])
    assert _eq(Ix, G["tiny_Ix0"]) and _eq(Iy, G["tiny_Iy0"]) and _eq(Ik, G["tiny_Ik0"])
    assert _eq(oracle.hog_features(synth.synthetic_frame(2, 61, 77, 1), 4), G["grey_feat"])
    o, ix, iy = oracle.dt(G["dt_in"], float(-np.float32(0.012)), float(-np.float32(0.004)), float(-np.float32(0.02)),
                          float(-np.float32(-0.007)), -3, 2)
    assert _eq(o, G["dt_out"]) and _eq(ix, G["dt_Ix"]) and _eq(iy, G["dt_Iy"])
    for name, model, frame in (("tiny", M.synthetic_tiny_model(thresh=0.7), im),
                               ("person", M.synthetic_person_model(thresh=17.9), synth.synthetic_frame(21, 160, 120, 3))):
        c = oracle.detect(model.flatten(), frame)
        hdr = np.array([(x["level"], x["component"], x["root_y"], x["root_x"]) for x in c], np.int32).reshape(-1, 4)
        assert _eq(hdr, G[f"{name}_cand_hdr"])
        assert _eq(np.array([x["score"] for x in c], np.float32), G[f"{name}_cand_score"])
        for i, x in enumerate(c):
            assert _eq(x["parts"], G[f"{name}_cand_parts"][i, :len(x["parts"])])


@pytest.mark.gpu
def test_hip_matches_golden():
    from partsbaseddetector_amd import detector as D
    model = M.synthetic_tiny_model(thresh=0.7)
    det = D.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(1, 96, 80, 3)
    feats = det.features_.pyramid(im)
    assert _eq(det.features_.scales(), G["tiny_scales"])
    imgs = det.features_.level_images(96, 80, 3)
    assert _eq(imgs[1], G["tiny_img1"]) and _eq(imgs[-1], G["tiny_img_last"])
    assert _eq(feats[0], G["tiny_feat0"]) and _eq(feats[-1], G["tiny_feat_last"])
    resp0 = det.convolution_engine_.pdf([feats[0]])[0]
    assert _eq(resp0, G["tiny_resp0"])
    Ix, Iy, Ik, rootv, rooti = det.dp_.min([resp0])
    assert _eq(rootv[0][0], G["tiny_rootv0"]) and _eq(rooti[0][0], G["tiny_rooti0"])
    assert _eq(Ix[0], G["tiny_Ix0"]) and _eq(Iy[0], G["tiny_Iy0"]) and _eq(Ik[0], G["tiny_Ik0"])
    grey = D.HOGFeatures(det.hd).pyramid(synth.synthetic_frame(2, 61, 77, 1))[0]
    assert _eq(grey, G["grey_feat"])
    for name, mdl, frame in (("tiny", model, im),
                             ("person", M.synthetic_person_model(thresh=17.9), synth.synthetic_frame(21, 160, 120, 3))):
        d2 = D.PartsBasedDetector(device=0)
        d2.distributeModel(mdl)
        c = d2.detect(frame)
        hdr = np.array([(x.level, x.component, x.root[1], x.root[0]) for x in c], np.int32).reshape(-1, 4)
        assert _eq(hdr, G[f"{name}_cand_hdr"])
        assert _eq(np.array([x.score() for x in c], np.float32), G[f"{name}_cand_score"])
        for i, x in enumerate(c):
            assert _eq(x.parts, G[f"{name}_cand_parts"][i, :len(x.parts)])
        d2.hd.close()
    det.hd.close()
