"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bars: integer/index outputs and (in EXACT convolution mode) float outputs
bit-exact; FMA convolution mode within 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest

from partsbaseddetector_amd import synth
from partsbaseddetector_amd import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det_mod():
    from partsbaseddetector_amd import detector
    return detector


def _handle(det_mod, model, **kw):
    return det_mod.Handle(model, device=0, **kw)


def _cand_key(c):
    return (c["level"], c["component"], c["root_y"], c["root_x"])


def _compare_candidates(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert (g.level, g.component, g.root[1], g.root[0]) == _cand_key(w)
        assert np.array_equal(g.parts, w["parts"]), (g.parts, w["parts"])
        assert np.float32(g.score()) == np.float32(w["score"])


@pytest.mark.parametrize("shape,cn", [((96, 80), 3), ((123, 157), 3), ((97, 131), 1), ((240, 320), 3)])
def test_pyramid_images_and_features(det_mod, oracle, shape, cn):
    model = M.synthetic_tiny_model()
    flat = model.flatten()
    hd = _handle(det_mod, flat)
    feats_eng = det_mod.HOGFeatures(hd)
    im = synth.synthetic_frame(11, shape[0], shape[1], cn)
    got = feats_eng.pyramid(im)
    want, scales = oracle.features_pyramid(flat, im)
    assert len(got) == len(want) == feats_eng.nscales()
    assert np.array_equal(feats_eng.scales(), scales)
    imgs_want, _ = oracle.pyramid_images(im, flat.sbin, flat.interval)
    imgs_got = feats_eng.level_images(shape[0], shape[1], cn)
    for l, (a, b) in enumerate(zip(imgs_got, imgs_want)):
        assert a.shape == b.shape, (l, a.shape, b.shape)
        assert np.array_equal(a, b), f"pyramid image level {l}: {np.count_nonzero(a != b)} pixels differ"
    for l, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape, (l, a.shape, b.shape)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            f"features level {l}: {np.count_nonzero(a != b)} of {a.size} differ, max {np.abs(a - b).max()}"
    hd.close()


def test_features_noise_and_constant_frames(det_mod, oracle):
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat)
    eng = det_mod.HOGFeatures(hd)
    for kind in ("noise", "constant"):
        im = synth.synthetic_frame(3, 90, 110, 3, kind=kind)
        got = eng.pyramid(im)
        want, _ = oracle.features_pyramid(flat, im)
        for a, b in zip(got, want):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if kind == "constant":
            assert all(not a.any() for a in got)   # zero gradients -> all 32 channels exactly 0
    hd.close()


def test_strided_image(det_mod, oracle):
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat)
    eng = det_mod.HOGFeatures(hd)
    big = synth.synthetic_frame(9, 120, 200, 3)
    view = big[:, 20:150]          # non-contiguous rows (cv::Mat ROI)
    got = eng.pyramid(view)
    want, _ = oracle.features_pyramid(flat, np.ascontiguousarray(view))
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    hd.close()


@pytest.mark.parametrize("mode", ["exact", "fma", "mfma"])
def test_conv_pdf(det_mod, oracle, mode):
    from partsbaseddetector_amd import _lib
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat, conv_mode={"exact": _lib.CONV_EXACT, "fma": _lib.CONV_FMA, "mfma": _lib.CONV_MFMA}[mode])
    conv = det_mod.SpatialConvolutionEngine(hd)
    rng = np.random.default_rng(5)
    # ragged levels incl. maps smaller than the filter and one empty level
    # (levels at least 64 cells wide are covered by tiles that wrap from one four-row strip into the next)
    dims = [(37, 45), (8, 33), (3, 2), (1, 1), (0, 5), (12, 70), (5, 64), (4, 65), (9, 130), (7, 201), (66, 97)]
    feats = [(rng.random((h, w * 32), dtype=np.float32) * 0.4) for h, w in dims]
    for f in feats:
        if f.size:
            f.reshape(f.shape[0], -1, 32)[:, :, 31] = 0.0
    feats[1].reshape(8, 33, 32)[:, :, 31] = 0.3   # a non-zero last channel inside the image
    feats[8].reshape(9, 130, 32)[:, :, 31] = 0.2
    got = conv.pdf(feats)
    for (h, w), f, g in zip(dims, feats, got):
        assert g.shape == (flat.nfilters, h, w)
        if h * w == 0:
            continue
        want = oracle.responses(flat, f)
        if mode == "exact":
            assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), \
                f"{h}x{w}: {np.count_nonzero(g != want)} differ, max {np.abs(g - want).max()}"
        else:
            assert np.abs(g - want).max() <= 1e-4
    hd.close()


def test_conv_set_filters_replaces_model_filters(det_mod, oracle):
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat)
    conv = det_mod.SpatialConvolutionEngine(hd)
    rng = np.random.default_rng(6)
    filters = [rng.standard_normal((5, 5 * 32)).astype(np.float32) * 0.1 for _ in range(11)]   # 11: not a multiple of 8
    conv.setFilters(filters)
    feat = rng.random((20, 41 * 32), dtype=np.float32)
    got = conv.pdf([feat])[0]
    assert got.shape == (11, 20, 41)
    for f in range(11):
        want = oracle.conv(feat, filters[f])
        assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32))
    hd.close()


@pytest.mark.parametrize("which", ["tiny", "tiny_plain", "tiny_wide", "face", "person_small"])
def test_dp_min(det_mod, oracle, which):
    if which in ("tiny", "tiny_wide"):
        model = M.synthetic_tiny_model(linear_def=True)
    elif which == "tiny_plain":
        model = M.synthetic_tiny_model(linear_def=False)
    elif which == "face":
        model = M.synthetic_face_model(nparts=9, ncomponents=2)
    else:
        model = M.synthetic_person_model()
    flat = model.flatten()
    hd = _handle(det_mod, flat)
    dp = det_mod.DynamicProgram(hd)
    rng = np.random.default_rng(8)
    dims = [(21, 30), (9, 7), (1, 6), (5, 1)] if which != "person_small" else [(30, 41), (6, 5)]
    if which == "tiny_wide":        # a side above 256 cells: the position planes are int16 instead of uint8
        dims = [(5, 300), (270, 4), (3, 256), (2, 257)]
    scores = [rng.standard_normal((flat.nfilters, h, w)).astype(np.float32) for h, w in dims]
    Ix, Iy, Ik, rootv, rooti = dp.min(scores)
    for l, s in enumerate(scores):
        for c in range(flat.ncomponents):
            oIx, oIy, oIk, orv, ori = oracle.dp_min(flat, c, s)
            assert np.array_equal(rootv[l][c].view(np.uint32), orv.view(np.uint32)), (l, c, np.abs(rootv[l][c] - orv).max())
            assert np.array_equal(rooti[l][c], ori)
            p0, p1 = flat.part_offset[c], flat.part_offset[c + 1]
            for gp in range(p0 + 1, p1):
                par = p0 + flat.parentid[gp]
                L = flat.mix_offset[par + 1] - flat.mix_offset[par]
                for m in range(L):
                    sl = flat.ptr_slot[gp] + m
                    assert np.array_equal(Ik[l][sl], oIk[sl]), (l, c, gp, m)
                    assert np.array_equal(Ix[l][sl], oIx[sl]), (l, c, gp, m)
                    assert np.array_equal(Iy[l][sl], oIy[sl]), (l, c, gp, m)
    hd.close()


@pytest.mark.parametrize("which,shape,thresh", [("tiny", (96, 128), 0.6), ("tiny", (150, 101), 0.2),
                                                 ("tiny", (44, 2200), 1.2),      # 548 cells wide: int16 position planes
                                                 ("face", (240, 320), None), ("person", (160, 120), None)])
def test_detect_end_to_end(det_mod, oracle, which, shape, thresh):
    if which == "tiny":
        model = M.synthetic_tiny_model(thresh=thresh)
    elif which == "face":
        model = M.synthetic_face_model(thresh=5.8)
    else:
        model = M.synthetic_person_model(thresh=17.9)
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(21, shape[0], shape[1], 3)
    got = det.detect(im)
    want = oracle.detect(flat, im)
    assert len(want) > 0
    _compare_candidates(got, want)
    # staged read-back of the same run: responses and root scores, bit-exact
    feats, _ = oracle.features_pyramid(flat, im)
    for l in (0, len(feats) - 1):
        H, W = feats[l].shape[0], feats[l].shape[1] // 32
        r = det.hd.get_stage(1, 0, l, H, W)
        want_r = oracle.responses(flat, feats[l])
        assert np.array_equal(r.view(np.uint32), want_r.view(np.uint32))
    det.hd.close()


def test_detect_batch_equals_single_calls(det_mod, oracle):
    model = M.synthetic_tiny_model(thresh=0.5)
    det = det_mod.PartsBasedDetector(device=0, max_batch=5)
    det.distributeModel(model)
    frames = [synth.synthetic_frame(30 + i, 100, 140, 3) for i in range(5)]
    batch = det.detect_batch(frames)
    off = 0
    for i, f in enumerate(frames):
        single = det.detect(f)
        mine = [c for c in batch if c.frame == i]
        assert len(mine) == len(single)
        for a, b in zip(mine, single):
            assert (a.level, a.component, a.root) == (b.level, b.component, b.root)
            assert np.array_equal(a.parts, b.parts) and a.score() == b.score()
        off += len(mine)
    assert off == len(batch)
    det.hd.close()


def test_errors(det_mod):
    from partsbaseddetector_amd._lib import PbdError
    model = M.synthetic_tiny_model(thresh=0.5)
    det = det_mod.PartsBasedDetector(device=0)
    with pytest.raises(PbdError):
        det.detect(np.zeros((100, 100, 3), np.uint8))          # before distributeModel
    det.distributeModel(model)
    with pytest.raises(PbdError):
        det.detect(np.zeros((10, 10, 3), np.uint8))            # too small for one octave
    with pytest.raises(PbdError):
        det.detect(np.zeros((100, 100, 3), np.int32))          # unsupported depth (CV_32S: src/HOGFeatures.cpp:141-145)
    with pytest.raises(PbdError):
        det.detect(np.zeros((100, 100, 2), np.uint8))          # channels must be 1 or 3
    with pytest.raises(PbdError) as e:
        det.detect(synth.synthetic_frame(1, 100, 100), capacity=1)   # capacity overflow is reported
    assert e.value.code == -4
    det.hd.close()


def test_allocation_failure_is_a_status_code(det_mod):
    """An allocation that cannot succeed (a candidate buffer of ~1 TB: 2^31 records of the person model's 448 bytes;
    the MI355X has 288 GB) comes back as PBD_ERR_NOMEM with a message,
    and the handle stays usable -- the reference reports errors as CV_Error / bool, never by dying
    (src/HOGFeatures.cpp:141-145, src/FileStorageModel.cpp:100-101)."""
    from partsbaseddetector_amd._lib import PbdError
    model = M.synthetic_person_model(thresh=17.9)
    im = synth.synthetic_frame(1, 120, 160)
    hd = det_mod.Handle(model, device=0, max_candidates=2 ** 31 - 1)
    buf = np.zeros(16 * hd.stride, np.int32)
    import ctypes as C
    n = C.c_int()
    rc = hd.lib.pbd_detect(hd.h, im.ctypes.data, 120, 160, 3, im.strides[0], buf.ctypes.data, 16, C.byref(n))
    assert rc == -6, (rc, hd.lib.pbd_last_error(hd.h))
    assert b"hipMalloc" in hd.lib.pbd_last_error(hd.h) or b"memory" in hd.lib.pbd_last_error(hd.h)
    hd.close()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    assert len(det.detect(im)) > 0
    det.hd.close()


def test_set_filters_that_do_not_cover_the_model(det_mod, oracle):
    """setFilters() with fewer filters than the model's filter ids: the convolution engine keeps working on its own,
    the model-dependent calls refuse (PBD_ERR_STATE) instead of indexing response planes out of bounds; installing a
    covering bank again restores them, with the part boxes following the new filter size."""
    from partsbaseddetector_amd._lib import PbdError
    model = M.synthetic_tiny_model(thresh=0.5)
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(1, 100, 100)
    rng = np.random.default_rng(1)
    det.convolution_engine_.setFilters([rng.standard_normal((5, 160)).astype(np.float32) for _ in range(flat.nfilters - 1)])
    feat = rng.random((9, 11 * 32), dtype=np.float32)
    assert det.convolution_engine_.pdf([feat])[0].shape == (flat.nfilters - 1, 9, 11)
    with pytest.raises(PbdError) as e:
        det.detect(im)
    assert e.value.code == -5
    filters = [flat.filters_f32[int(flat.filter_offset[f]):int(flat.filter_offset[f]) + 5 * 160].reshape(5, 160) for f in range(flat.nfilters)]
    det.convolution_engine_.setFilters(filters)
    _compare_candidates(det.detect(im), oracle.detect(flat, im))
    det.hd.close()


@pytest.mark.parametrize("shape,seed", [((480, 640), 1), ((1080, 1920), 2)])
def test_person_model_full_size_frames(det_mod, oracle, shape, seed):
    """BASELINE configs[1]/[3] sizes: one 640x480 and one 1920x1080 frame, whole path, vs the oracle."""
    model = M.synthetic_person_model(thresh=18.9 if shape[0] == 480 else 19.3)
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(seed, shape[0], shape[1], 3)
    got = det.detect(im)
    want = oracle.detect(flat, im)
    assert len(want) > 0
    _compare_candidates(got, want)
    det.hd.close()


def test_face_config_plumbing(det_mod, oracle, tmp_path):
    """BASELINE configs[0]: a `.by_parts` configuration (pipeline1.parameters.extra.model_file, as
    conf/config_face.by_parts:31) names an XML model that does not ship -> the synthetic face-like stand-in (many parts,
    1 mixture, several components sharing filters), written to and re-read from the XML flavour of the model format,
    one 320x240 frame, interval 5."""
    from partsbaseddetector_amd import config as CFG, filestorage as FS
    (tmp_path / "config_face.by_parts").write_text(
        "pipeline1:\n  type: PartsBasedDetector\n  parameters:\n    extra:\n      model_file: \"/nowhere/Face_68parts.xml\"\n      use_cuda: false\n")
    cfg = CFG.load_by_parts(str(tmp_path / "config_face.by_parts"))[0]
    stand_in = CFG.load_model(cfg, stand_in=M.synthetic_face_model(thresh=6.0, nparts=20, ncomponents=3, interval=5))
    FS.serialize_xml(stand_in, str(tmp_path / "Face_68parts.xml"))
    model = CFG.load_model(cfg, search_dirs=[str(tmp_path)])           # now found next to the config: the XML reader's output
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(7, 240, 320, 3)
    _compare_candidates(det.detect(im), oracle.detect(flat, im))
    assert det.features_.nscales() == 18       # 320x240, sbin 4, interval 5
    det.hd.close()


# ---------------------------------------------------------------------------------------------------
# T = double (the reference's ECTO / ROS instantiation, cells/detect.cpp:93, ros/Node.hpp:121)
# ---------------------------------------------------------------------------------------------------
def _eq64(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_f64_features_conv_dp(det_mod, oracle):
    from partsbaseddetector_amd import _lib
    model = M.synthetic_tiny_model(linear_def=True)
    flat = model.flatten()
    hd = det_mod.Handle(flat, device=0, real_type=_lib.REAL_F64)
    feats_eng = det_mod.HOGFeatures(hd)
    for shape, cn in (((123, 157), 3), ((97, 131), 1)):
        im = synth.synthetic_frame(11, shape[0], shape[1], cn)
        got = feats_eng.pyramid(im)
        want, scales = oracle.features_pyramid(flat, im, dtype=np.float64)
        assert np.array_equal(feats_eng.scales(), scales)
        for l, (a, b) in enumerate(zip(got, want)):
            assert a.dtype == np.float64 and _eq64(a, b), (l, np.abs(a - b).max())
    conv = det_mod.SpatialConvolutionEngine(hd)
    rng = np.random.default_rng(5)
    dims = [(37, 45), (3, 2), (12, 70)]
    feats = [rng.random((h, w * 32)) * 0.4 for h, w in dims]
    got = conv.pdf(feats)
    for f, g in zip(feats, got):
        assert _eq64(g, oracle.responses(flat, f))
    dp = det_mod.DynamicProgram(hd)
    scores = [rng.standard_normal((flat.nfilters, h, w)) for h, w in [(21, 30), (9, 7), (1, 6)]]
    Ix, Iy, Ik, rootv, rooti = dp.min(scores)
    for l, s in enumerate(scores):
        oIx, oIy, oIk, orv, ori = oracle.dp_min(flat, 0, s)
        assert _eq64(rootv[l][0], orv) and np.array_equal(rooti[l][0], ori)
        assert np.array_equal(Ix[l], oIx) and np.array_equal(Iy[l], oIy) and np.array_equal(Ik[l], oIk)
    hd.close()


@pytest.mark.parametrize("which,shape,thresh", [("tiny", (96, 128), 0.6), ("person", (160, 120), 17.9), ("person", (480, 640), 18.9)])
def test_f64_detect_end_to_end(det_mod, oracle, which, shape, thresh):
    model = M.synthetic_tiny_model(thresh=thresh) if which == "tiny" else M.synthetic_person_model(thresh=thresh)
    det = det_mod.PartsBasedDetector(device=0, dtype=np.float64)
    det.distributeModel(model)
    im = synth.synthetic_frame(21, shape[0], shape[1], 3)
    got = det.detect(im)
    want = oracle.detect(model.flatten(), im, dtype=np.float64)
    assert len(want) > 0
    _compare_candidates(got, want)
    det.hd.close()


def test_mfma_mode_person_model(det_mod, oracle):
    """PBD_CONV_MFMA on the 156-filter person model: responses within 1e-4 of the reference order (north-star
    tolerance); candidates compared with the exact path (identical here; not guaranteed on near ties)."""
    from partsbaseddetector_amd import _lib
    model = M.synthetic_person_model(thresh=17.9)
    flat = model.flatten()
    im = synth.synthetic_frame(21, 160, 120, 3)
    det = det_mod.PartsBasedDetector(device=0, conv_mode=_lib.CONV_MFMA)
    det.distributeModel(model)
    got = det.detect(im)
    feats, _ = oracle.features_pyramid(flat, im)
    worst = 0.0
    for l in (0, 3, len(feats) - 1):
        H, W = feats[l].shape[0], feats[l].shape[1] // 32
        r = det.hd.get_stage(1, 0, l, H, W)
        worst = max(worst, float(np.abs(r - oracle.responses(flat, feats[l])).max()))
    assert worst <= 1e-4, worst
    want = oracle.detect(flat, im)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (g.level, g.component, g.root[1], g.root[0]) == _cand_key(w)
        assert np.array_equal(g.parts, w["parts"]) and abs(g.score() - w["score"]) <= 1e-4
    det.hd.close()


def test_mfma_more_than_one_filter_block(det_mod, oracle):
    """PBD_CONV_MFMA sweeps the filters in blocks of 160: 170 filters need two passes."""
    from partsbaseddetector_amd import _lib
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat, conv_mode=_lib.CONV_MFMA)
    conv = det_mod.SpatialConvolutionEngine(hd)
    rng = np.random.default_rng(12)
    filters = [rng.standard_normal((5, 5 * 32)).astype(np.float32) * 0.05 for _ in range(170)]
    conv.setFilters(filters)
    feat = rng.random((19, 37 * 32), dtype=np.float32) * 0.4
    got = conv.pdf([feat])[0]
    assert got.shape == (170, 19, 37)
    for f in (0, 31, 159, 160, 169):
        assert np.abs(got[f] - oracle.conv(feat, filters[f])).max() <= 1e-4
    hd.close()


def test_no_candidates_and_alternating_sizes(det_mod, oracle):
    """thresh above every root score -> empty list (not an error); frames of different sizes through one
    handle (plan cache, grow-only workspace) give the same answers as fresh handles."""
    model = M.synthetic_tiny_model(thresh=1e9)
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    assert det.detect(synth.synthetic_frame(1, 96, 128, 3)) == []
    det.hd.close()
    model = M.synthetic_tiny_model(thresh=0.8)
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    frames = [synth.synthetic_frame(3, 96, 128, 3), synth.synthetic_frame(4, 211, 173, 3), synth.synthetic_frame(5, 64, 300, 1),
              synth.synthetic_frame(3, 96, 128, 3)]
    for f in frames:
        _compare_candidates(det.detect(f), oracle.detect(flat, f))
    det.hd.close()


def test_mfma_f16_mode(det_mod, oracle):
    """PBD_CONV_MFMA_F16 (BASELINE.json configs[4] / SURVEY.md 8(d) config 5): operands rounded once to fp16, one
    MFMA per product tile, fp32 accumulation.  The 1e-4 score bar does not apply; checked instead:
      * against the reference-order convolution of the SAME fp16-rounded operands (numpy float16 rounding):
        only the fp32 summation order differs -> 2e-5;
      * against the unrounded reference: max-abs error reported, bounded by 5e-3;
      * detections vs the exact path on the person model: agreement rate reported, root set overlap >= 90 %."""
    from partsbaseddetector_amd import _lib
    flat = M.synthetic_tiny_model().flatten()
    hd = _handle(det_mod, flat, conv_mode=_lib.CONV_MFMA_F16)
    conv = det_mod.SpatialConvolutionEngine(hd)
    rng = np.random.default_rng(5)
    filters = [rng.standard_normal((5, 5 * 32)).astype(np.float32) * 0.05 for _ in range(170)]   # two filter blocks
    filters[3][2, 7] = 1e-7      # fp16 subnormal range
    filters[4][1, 9] = 3e-9      # below half the smallest subnormal -> 0
    conv.setFilters(filters)
    feat = rng.random((19, 37 * 32), dtype=np.float32) * 0.4
    got = conv.pdf([feat])[0]
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    worst_same, worst_ref = 0.0, 0.0
    for f in (0, 3, 4, 31, 159, 160, 169):
        same = oracle.conv(f16(feat), f16(filters[f]))
        # the responses themselves are fp16 in this mode (BASELINE configs[4]): one more rounding of 2^-11 relative
        assert np.all(np.abs(got[f] - same) <= 2e-5 + 2.0 ** -10 * np.abs(same)), f
        assert np.array_equal(got[f], f16(got[f]))                      # exactly representable halves
        worst_same = max(worst_same, float(np.abs(got[f] - f16(same)).max()))
        worst_ref = max(worst_ref, float(np.abs(got[f] - oracle.conv(feat, filters[f])).max()))
    print(f"fp16 mode: max |resp - fp16(reference(fp16 operands))| = {worst_same:.3g}, max |resp - reference| = {worst_ref:.3g}")
    assert worst_ref <= 5e-3, worst_ref
    hd.close()

    model = M.synthetic_person_model(thresh=17.9)
    im = synth.synthetic_frame(21, 160, 120, 3)
    res = {}
    for name, mode in (("exact", _lib.CONV_EXACT), ("f16", _lib.CONV_MFMA_F16)):
        det = det_mod.PartsBasedDetector(device=0, conv_mode=mode)
        det.distributeModel(model)
        res[name] = {(c.level, c.component, c.root[1], c.root[0]): c for c in det.detect(im)}
        det.hd.close()
    common = set(res["exact"]) & set(res["f16"])
    same_parts = sum(np.array_equal(res["exact"][k].parts, res["f16"][k].parts) for k in common)
    dscore = max((abs(res["exact"][k].score() - res["f16"][k].score()) for k in common), default=0.0)
    print(f"fp16 mode detections: exact {len(res['exact'])}, fp16 {len(res['f16'])}, common roots {len(common)}, "
          f"identical part placements {same_parts}, max score diff {dscore:.3g}")
    assert len(res["exact"]) > 0
    assert len(common) >= 0.9 * max(len(res["exact"]), len(res["f16"]))
    assert dscore <= 5e-2


@pytest.mark.parametrize("ksize", [3, 7])
def test_training_demo_configuration(det_mod, oracle, ksize):
    """The reference's in-tree example model geometry (matlab/training_demo.m:5,12,29: K = [4 4 4 4 4 4],
    pa = [0 1 1 3 2 4], sbin = 8) with 3x3 / 7x7 filters: sbin 8 HOG, the generic-size convolution kernel and a
    6-part tree, every stage bit for bit against the oracle."""
    model = M.synthetic_model(seed=11, pa=[0, 1, 1, 3, 2, 4], nmix=4, ksize=ksize, sbin=8, interval=5, thresh=-1e9,
                              name="training-demo")
    flat = model.flatten()
    im = synth.synthetic_frame(33, 200, 264, 3)
    want = oracle.detect(flat, im)
    model.thresh = float(np.sort([w["score"] for w in want])[-40])      # keep the 40 best roots
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    got = det.detect(im)
    want = oracle.detect(flat, im)
    assert len(want) >= 30
    _compare_candidates(got, want)
    feats, _ = oracle.features_pyramid(flat, im)
    for l in (0, len(feats) - 1):
        H, W = feats[l].shape[0], feats[l].shape[1] // 32
        assert np.array_equal(det.hd.get_stage(0, 0, l, H, W).view(np.uint32), feats[l].view(np.uint32))
        assert np.array_equal(det.hd.get_stage(1, 0, l, H, W).view(np.uint32), oracle.responses(flat, feats[l]).view(np.uint32))
    det.hd.close()


def test_config2_gpu_features_conv_host_dp(det_mod, oracle):
    """BASELINE configs[1]: person model, one 640x480 frame, HOG + convolution on the GPU through the staged
    IFeatures / IConvolutionEngine surface, responses back on the host, DynamicProgram::min/argmin on the host
    (CPU restatement).  The candidates must equal the all-CPU and the all-GPU results."""
    model = M.synthetic_person_model()
    flat = model.flatten()
    im = synth.synthetic_frame(2, 480, 640, 3)
    hd = _handle(det_mod, flat)
    fe = det_mod.HOGFeatures(hd)
    feats = fe.pyramid(im)
    resp = det_mod.SpatialConvolutionEngine(hd).pdf(feats)       # responses[level] = (nfilters, H, W) on the host
    scales = fe.scales()
    hd.close()
    assert len(resp) == 46 and sum(r.shape[1] * r.shape[2] for r in resp) == 140725
    host = []
    for l, r in enumerate(resp):
        if r.shape[1] == 0 or r.shape[2] == 0:
            continue
        for c in range(flat.ncomponents):
            Ix, Iy, Ik, rootv, rooti = oracle.dp_min(flat, c, r)
            host += oracle.dp_argmin(flat, c, l, float(scales[l]), Ix, Iy, Ik, rootv, rooti)
    host.sort(key=lambda w: (w["level"], w["component"], w["root_y"], w["root_x"]))
    want = oracle.detect(flat, im)
    assert len(want) > 0 and len(host) == len(want)
    for a, b in zip(host, want):
        assert _cand_key(a) == _cand_key(b) and np.array_equal(a["parts"], b["parts"])
        assert np.float32(a["score"]) == np.float32(b["score"])
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    _compare_candidates(det.detect(im), want)
    det.hd.close()


def test_pipelined_submit_wait_equals_synchronous_batches(det_mod, oracle):
    """pbd_detect_batch_submit / _wait with two batches in flight: every batch's records equal those of the synchronous
    pbd_detect_batch on the same frames (and, for one frame, the oracle); call-order violations are status codes."""
    from partsbaseddetector_amd._lib import PbdError
    model = M.synthetic_tiny_model(thresh=0.5)
    flat = model.flatten()
    det = det_mod.PartsBasedDetector(device=0, max_batch=4)
    det.distributeModel(model)
    batches = [[synth.synthetic_frame(100 + 10 * b + i, 120, 150, 3) for i in range(4 if b != 2 else 3)] for b in range(5)]
    want = [[(c.frame, c.level, c.component, c.root, c.score(), c.parts.tobytes()) for c in det.detect_batch(fr)] for fr in batches]
    with pytest.raises(PbdError) as e:
        det.wait_batch()                                   # nothing in flight
    assert e.value.code == -5
    got = []
    det.submit_batch(batches[0])
    for b in range(1, len(batches)):
        det.submit_batch(batches[b])                       # two in flight
        if b == 1:
            with pytest.raises(PbdError) as e:
                det.submit_batch(batches[0])               # a third is refused
            assert e.value.code == -5
            with pytest.raises(PbdError) as e:
                det.detect(batches[0][0])                  # and so is a synchronous call
            assert e.value.code == -5
        got.append(det.wait_batch())
    got.append(det.wait_batch())
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert [(c.frame, c.level, c.component, c.root, c.score(), c.parts.tobytes()) for c in g] == w
    _compare_candidates([c for c in got[3] if c.frame == 2], oracle.detect(flat, batches[3][2]))
    assert len(det.detect(batches[0][0])) > 0              # synchronous calls work again
    det.hd.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_model_with_filters_of_different_sizes(det_mod, oracle, dtype):
    """Filters of several sizes inside one model: the reference builds one engine per filter and sizes the part boxes
    per mixture (src/SpatialConvolutionEngine.cpp:141-158, include/Parts.hpp:185-187).  Sizes 5 / 3 / 7 / 4 cycling over
    the 12 filters of a 4-part x 3-mixture tree (even size: anchor k/2): responses of every filter, the candidates and
    their per-mixture box sizes bit for bit against the oracle."""
    model = M.synthetic_model(seed=17, pa=[0, 1, 1, 2], nmix=3, ksize=[5, 3, 7, 4], interval=5, thresh=-1e9, name="mixed-sizes")
    flat = model.flatten()
    assert sorted(set(int(k) for k in flat.filter_ksize)) == [3, 4, 5, 7]
    im = synth.synthetic_frame(41, 150, 130, 3)
    want = oracle.detect(flat, im, dtype=dtype)
    model.thresh = float(np.sort([w["score"] for w in want])[-60])
    flat = model.flatten()
    want = oracle.detect(flat, im, dtype=dtype)
    det = det_mod.PartsBasedDetector(device=0, dtype=dtype)
    det.distributeModel(model)
    got = det.detect(im)
    _compare_candidates(got, want)
    assert len({tuple(p[2:]) for c in got for p in c.parts.tolist()}) > 4          # several box sizes occur
    feats, _ = oracle.features_pyramid(flat, im, dtype=dtype)
    for l in (0, len(feats) - 1):
        H, W = feats[l].shape[0], feats[l].shape[1] // 32
        r = det.hd.get_stage(1, 0, l, H, W)
        wr = oracle.responses(flat, feats[l])
        assert r.dtype == wr.dtype and np.array_equal(r.view(np.uint8), wr.view(np.uint8))
    det.hd.close()


@pytest.mark.parametrize("dtype,ksizes,nmix", [(np.float32, [9, 12, 5], 3), (np.float64, [9, 8], 2), (np.float32, [5], 12), (np.float64, [3], 10)])
def test_large_filters_and_many_mixtures(det_mod, oracle, dtype, ksizes, nmix):
    """The reference sizes nothing at compile time (include/Parts.hpp:51-261, src/SpatialConvolutionEngine.cpp:141-158).  Filters
    larger than 7 x 7 run on the generic kernel -- with the haloed tile staged a few channels at a time when 32 channels do
    not fit LDS (T = double from 9 x 9 on) -- and parts with more than 8 mixtures on the 16-wide combine / root kernels:
    responses of every filter size and all candidates bit for bit against the oracle."""
    model = M.synthetic_model(seed=29 + nmix, pa=[0, 1, 1, 2], nmix=nmix, ksize=ksizes, interval=5, thresh=-1e9, name="large")
    flat = model.flatten()
    assert max(int(k) for k in flat.filter_ksize) == max(ksizes)
    im = synth.synthetic_frame(43, 140, 120, 3)
    want = oracle.detect(flat, im, dtype=dtype)
    model.thresh = float(np.sort([w["score"] for w in want])[-50])
    flat = model.flatten()
    want = oracle.detect(flat, im, dtype=dtype)
    det = det_mod.PartsBasedDetector(device=0, dtype=dtype)
    det.distributeModel(model)
    got = det.detect(im)
    _compare_candidates(got, want)
    feats, _ = oracle.features_pyramid(flat, im, dtype=dtype)
    for l in (0, len(feats) - 1):
        H, W = feats[l].shape[0], feats[l].shape[1] // 32
        r = det.hd.get_stage(1, 0, l, H, W)
        wr = oracle.responses(flat, feats[l])
        assert r.dtype == wr.dtype and np.array_equal(r.view(np.uint8), wr.view(np.uint8))
    det.hd.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_filter_shared_inside_a_component(det_mod, oracle, dtype):
    """A filter id referenced by more than one (part, mixture) of a component: the reference keys the accumulated
    scores by filter id (src/DynamicProgram.cpp:93,115-119,154-156; include/Parts.hpp:168-171), so such parts see each
    other's children's messages in processing order.  Cases: two siblings sharing both filters, a child sharing a filter
    with its own parent, two mixtures of one part pointing at the same filter -- rootv / rooti / Ix / Iy / Ik and the
    candidates bit for bit against the oracle (which keeps `ncscores` exactly as the reference does)."""
    from partsbaseddetector_amd import _lib
    rng = np.random.default_rng(31)
    for case in range(3):
        model = M.synthetic_model(seed=23 + case, pa=[0, 1, 1, 2, 2, 3], nmix=2, interval=5, thresh=0.0, linear_def=True, name=f"shared-{case}")
        fid = model.filterid[0]
        if case == 0:
            fid[2] = list(fid[1])                      # siblings 1 and 2 (children of the root) share both filters
        elif case == 1:
            fid[3][0] = fid[1][1]                      # part 3 shares a filter with its parent (part 1) ...
            fid[5] = [fid[2][0], fid[2][0]]            # ... and part 5 uses its parent's (part 2) filter for both mixtures
        else:
            fid[0][1] = fid[0][0]                      # both root mixtures on one filter
            fid[4] = list(fid[3])                      # siblings 3 and 4 share
            fid[5][1] = fid[1][0]                      # a grandchild's filter = its grandparent's
        flat = model.flatten()
        hd = det_mod.Handle(flat, device=0, real_type=_lib.REAL_F32 if dtype == np.float32 else _lib.REAL_F64)
        dp = det_mod.DynamicProgram(hd)
        scores = [rng.standard_normal((flat.nfilters, h, w)).astype(dtype) for h, w in [(17, 23), (6, 9), (1, 5)]]
        Ix, Iy, Ik, rootv, rooti = dp.min(scores)
        for l, sc in enumerate(scores):
            oIx, oIy, oIk, orv, ori = oracle.dp_min(flat, 0, sc)
            assert np.array_equal(rootv[l][0].view(np.uint8), orv.view(np.uint8)), (case, l)
            assert np.array_equal(rooti[l][0], ori), (case, l)
            assert np.array_equal(Ix[l], oIx) and np.array_equal(Iy[l], oIy) and np.array_equal(Ik[l], oIk), (case, l)
        hd.close()
        im = synth.synthetic_frame(50 + case, 120, 100, 3)
        want = oracle.detect(flat, im, dtype=dtype)
        model.thresh = float(np.sort([w["score"] for w in want])[-30])
        det = det_mod.PartsBasedDetector(device=0, dtype=dtype)
        det.distributeModel(model)
        _compare_candidates(det.detect(im), oracle.detect(model.flatten(), im, dtype=dtype))
        det.hd.close()


@pytest.mark.parametrize("IT", [np.uint16, np.float32, np.float64])
def test_image_depths_16u_32f_64f(det_mod, oracle, IT):
    """HOGFeatures::pyramid on the other depths it accepts (src/HOGFeatures.cpp:136-146: features<uint16_t|float|double>),
    for T = float and T = double, colour and grey: level images, features of every level and the end-to-end candidates
    bit for bit against the oracle (the resampling of these depths is third-party arithmetic, restated: unpinned)."""
    from partsbaseddetector_amd import _lib
    rng = np.random.default_rng(11)
    for cn in (3, 1):
        base = synth.synthetic_frame(15, 110, 150, cn).astype(np.float64)
        if cn == 1:
            base = base.reshape(110, 150, 1)
        if IT == np.uint16:
            im = (base * 257 + rng.integers(0, 200, base.shape)).astype(np.uint16)      # genuine 16-bit range
        else:
            im = (base / 255.0 + rng.random(base.shape) * 1e-3).astype(IT)            # [0, 1] floats, as a float image pipeline would hand over
        for T in (np.float32, np.float64):
            model = M.synthetic_tiny_model(thresh=-1e9)
            flat = model.flatten()
            hd = det_mod.Handle(flat, device=0, real_type=_lib.REAL_F32 if T == np.float32 else _lib.REAL_F64)
            eng = det_mod.HOGFeatures(hd)
            got = eng.pyramid(im)
            want, scales = oracle.features_pyramid(flat, im, dtype=T)
            assert len(got) == len(want) and np.array_equal(eng.scales(), scales)
            imgs_want, _ = oracle.pyramid_images(im, flat.sbin, flat.interval)
            imgs_got = eng.level_images(110, 150, cn, IT)
            for l, (a, b) in enumerate(zip(imgs_got, imgs_want)):
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (cn, T, "image level", l)
            for l, (a, b) in enumerate(zip(got, want)):
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (cn, T, "features level", l)
            hd.close()
            cands = oracle.detect(flat, im, dtype=T)
            model.thresh = float(np.sort([w["score"] for w in cands])[-25])
            det = det_mod.PartsBasedDetector(device=0, dtype=T)
            det.distributeModel(model)
            _compare_candidates(det.detect(im), oracle.detect(model.flatten(), im, dtype=T))
            det.hd.close()


def test_level_sharding_of_one_frame(det_mod, oracle):
    """SURVEY 8(e) secondary partitioning: one frame split over `world` GPUs by pyramid level.  Rehearsed on one GPU:
    the handle takes the role of every rank in turn; the union of the per-rank candidate lists is the full (oracle)
    result, every level belongs to exactly one rank, and the longest-processing-time assignment balances the cell
    counts (level 0 alone is 13 % of a VGA pyramid)."""
    model = M.synthetic_person_model(thresh=18.3)
    flat = model.flatten()
    im = synth.synthetic_frame(3, 480, 640, 3)
    want = oracle.detect(flat, im)
    assert len(want) > 20
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    full_plan = det.hd.plan(480, 640)
    for world in (2, 4):
        got, loads, owned = [], [], np.zeros(full_plan["nlevels"], np.int32)
        for rank in range(world):
            det.hd.set_level_shard(rank, world)
            plan = det.hd.plan(480, 640)
            cells = plan["feat_rows"].astype(np.int64) * plan["feat_cols"]
            owned += (cells > 0)
            loads.append(int(cells.sum()))
            mine = det.detect(im)
            assert all(cells[c.level] > 0 for c in mine)
            got += mine
        full_cells = full_plan["feat_rows"].astype(np.int64) * full_plan["feat_cols"]
        assert np.array_equal(owned, (full_cells > 0).astype(np.int32))          # a partition of the levels
        assert sum(loads) == int(full_cells.sum()) and max(loads) <= 1.1 * sum(loads) / world, loads
        got.sort(key=lambda c: (c.level, c.component, c.root[1], c.root[0]))
        _compare_candidates(got, want)
    det.hd.set_level_shard(0, 1)
    _compare_candidates(det.detect(im), want)
    det.hd.close()


def test_failed_set_filters_keeps_the_old_bank(det_mod, oracle):
    """setFilters() used to release the old bank before the new one was complete, so a failure (here: 3x3
    filters on a matrix-core handle -> PBD_ERR_UNSUPPORTED) left the handle half-updated.  The new bank is now built beside
    the old one and swapped in only on success: after the failed call the old filters still answer, bit for bit."""
    from partsbaseddetector_amd import _lib
    from partsbaseddetector_amd._lib import PbdError
    flat = M.synthetic_tiny_model().flatten()
    rng = np.random.default_rng(8)
    feat = rng.random((17, 29 * 32), dtype=np.float32) * 0.4
    for mode in (_lib.CONV_MFMA, _lib.CONV_EXACT):
        hd = _handle(det_mod, flat, conv_mode=mode)
        conv = det_mod.SpatialConvolutionEngine(hd)
        before = conv.pdf([feat])[0]
        bad = [rng.standard_normal((3, 3 * 32)).astype(np.float32) for _ in range(4)] if mode == _lib.CONV_MFMA else \
              [rng.standard_normal((33, 33 * 32)).astype(np.float32) for _ in range(4)]          # 33x33: outside 1..31 in every mode
        with pytest.raises(PbdError) as e:
            conv.setFilters(bad)
        assert e.value.code == -2
        conv._nfilters = flat.nfilters
        after = conv.pdf([feat])[0]
        assert np.array_equal(before.view(np.uint32), after.view(np.uint32))
        if mode == _lib.CONV_EXACT:
            assert np.array_equal(after.view(np.uint32), oracle.responses(flat, feat).view(np.uint32))
        # and the detector built on the handle still works
        hd.close()


def test_mfma_f16_dp_min_rounds_its_input(det_mod, oracle):
    """PBD_CONV_MFMA_F16 keeps the responses as fp16 on the device (include/pbd.h), so the staged pbd_dp_min rounds the CALLER's
    float scores to fp16 first: its result equals the oracle's dynamic program run on the fp16-rounded scores (bit for bit),
    not on the unrounded ones; scores beyond +-65504 saturate to inf."""
    from partsbaseddetector_amd import _lib
    model = M.synthetic_tiny_model()
    flat = model.flatten()
    hd = _handle(det_mod, flat, conv_mode=_lib.CONV_MFMA_F16)
    dp = det_mod.DynamicProgram(hd)
    rng = np.random.default_rng(12)
    dims = [(21, 30), (9, 14)]
    scores = [rng.standard_normal((flat.nfilters, h, w)).astype(np.float32) * 1.7 for h, w in dims]      # not fp16-representable
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    assert not np.array_equal(scores[0], f16(scores[0]))
    Ix, Iy, Ik, rootv, rooti = dp.min(scores)
    for l in range(len(dims)):
        for c in range(flat.ncomponents):
            oIx, oIy, oIk, orv, ori = oracle.dp_min(flat, c, f16(scores[l]))
            assert np.array_equal(rootv[l][c].view(np.uint32), orv.view(np.uint32))
            assert np.array_equal(rooti[l][c], ori)
            p0, p1 = flat.part_offset[c], flat.part_offset[c + 1]
            for gp in range(p0 + 1, p1):
                par = p0 + flat.parentid[gp]
                for m in range(flat.mix_offset[par + 1] - flat.mix_offset[par]):
                    sl = flat.ptr_slot[gp] + m
                    assert np.array_equal(Ik[l][sl], oIk[sl]) and np.array_equal(Ix[l][sl], oIx[sl]) and np.array_equal(Iy[l][sl], oIy[sl])
            unrounded = oracle.dp_min(flat, c, scores[l])[3]
            assert not np.array_equal(rootv[l][c], unrounded)
    hd.close()


@pytest.mark.parametrize("IT", [np.float32, np.float64])
def test_non_finite_pixels_are_refused(det_mod, IT):
    """32F / 64F images may carry NaN / Inf; the reference computes something deterministic from them, this library defines
    them as an error (PBD_ERR_INVALID) rather than risking a silently different envelope walk."""
    from partsbaseddetector_amd._lib import PbdError
    model = M.synthetic_tiny_model(thresh=0.5)
    det = det_mod.PartsBasedDetector(device=0)
    det.distributeModel(model)
    im = synth.synthetic_frame(3, 96, 128, 3).astype(IT)
    good = det.detect(im)
    for bad_value in (np.nan, np.inf, -np.inf):
        bad = im.copy()
        bad[40, 77, 1] = bad_value
        with pytest.raises(PbdError) as e:
            det.detect(bad)
        assert e.value.code == -1 and "NaN or Inf" in str(e.value)
        with pytest.raises(PbdError):
            det.features_.pyramid(bad)
    again = det.detect(im)                                   # the handle is still usable
    assert [(c.level, c.root, c.score()) for c in again] == [(c.level, c.root, c.score()) for c in good]
    det.hd.close()


def test_fused_path_responses_every_level(det_mod, oracle):
    """The fused detect path runs the exact convolution WITHOUT channel 31 (this library's HOG writes +0 there) and adds, for
    windows that leave the image, the tabulated ordered sum of the out-of-image taps' channel-31 weights (border value 1:
    src/SpatialConvolutionEngine.cpp:147-156) as the last term.  Every response of every level -- down to maps of 5 x 4 cells,
    where a window sticks out on both sides at once -- must equal the reference-order convolution of the same features."""
    from partsbaseddetector_amd import _lib
    model = M.synthetic_person_model(thresh=1e9)            # no candidates: only the staged responses matter
    flat = model.flatten()
    for shape in ((120, 161), (97, 83)):
        im = synth.synthetic_frame(77, shape[0], shape[1], 3)
        det = det_mod.PartsBasedDetector(device=0)
        det.distributeModel(model)
        assert det.detect(im) == []
        plan = det.hd.plan(*shape)
        feats, _ = oracle.features_pyramid(flat, im)
        checked = 0
        for l in range(plan["nlevels"]):
            h, w = int(plan["feat_rows"][l]), int(plan["feat_cols"][l])
            if h * w == 0:
                continue
            got = det.hd.get_stage(_lib.STAGE_RESPONSES, 0, l, h, w)
            f = det.hd.get_stage(_lib.STAGE_FEATURES, 0, l, h, w)
            assert np.array_equal(f.view(np.uint32), np.ascontiguousarray(feats[l], np.float32).reshape(f.shape).view(np.uint32))
            assert not f.reshape(h, w, 32)[:, :, 31].any()
            want = oracle.responses(flat, f)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (shape, l, h, w, np.abs(got - want).max())
            checked += 1
        assert checked >= 10 and min(int(plan["feat_rows"][-1]), int(plan["feat_cols"][-1])) <= 5
        det.hd.close()


def test_narrow_wave_distance_transform(det_mod, oracle):
    """Launches of the distance-transform passes that do not fill the chip run as more, narrower waves (64 >> lane_shift rows
    per wave, chosen per launch: pbd_kernels_dp.hip, dt_lane_shift).  The rows' arithmetic is untouched, so every setting must
    give the oracle's candidates: each of 64 / 32 / 16 / 8 / 4 / 2 / 1 rows per wave is forced in a fresh interpreter
    (PBD_DT_LANESHIFT is read once per process) on a frame with rows of up to 78 cells and int16-free uint8 planes, and once
    on a frame wide enough for int16 planes."""
    import hashlib, json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from partsbaseddetector_amd import detector, synth, model as M\n"
        "out = {}\n"
        "for name, shape, sbin in (('u8', (150, 330), 4), ('i16', (60, 1100), 4)):\n"
        "    model = M.synthetic_model(seed=31, pa=[0, 1, 1, 2, 2, 3], nmix=3, sbin=sbin, interval=4, thresh=-0.35, linear_def=(name == 'i16'), name='narrow')\n"
        "    det = detector.PartsBasedDetector(device=0)\n"
        "    det.distributeModel(model)\n"
        "    cands = det.detect(synth.synthetic_frame(47, shape[0], shape[1], 3))\n"
        "    h = hashlib.sha256()\n"
        "    for c in cands:\n"
        "        h.update(np.asarray([c.level, c.component, c.root[0], c.root[1]], np.int32).tobytes()); h.update(np.float32(c.score()).tobytes()); h.update(np.ascontiguousarray(c.parts, dtype=np.int32).tobytes())\n"
        "    out[name] = [len(cands), h.hexdigest()]\n"
        "    det.hd.close()\n"
        "print(json.dumps(out))\n" % root)
    results = {}
    # launches with eight or fewer rows per wave (lane_shift >= 3) go to the wavefront-cooperative kernel k_dt_coop (four rows
    # per wave, sixteen lanes each holding the envelope's top block) unless PBD_DT_COOP=0: both forms are forced
    for shift, coop in [("auto", None), ("0", None), ("1", None), ("2", None), ("3", None), ("4", None), ("5", None), ("6", None),
                        ("3", "0"), ("4", "0"), ("6", "0"), ("4", "g8")]:
        env = dict(os.environ)
        for k_ in ("PBD_DT_LANESHIFT", "PBD_DT_COOP", "PBD_DT_COOP_G"):
            env.pop(k_, None)
        if shift != "auto":
            env["PBD_DT_LANESHIFT"] = shift
        if coop == "g8":
            env["PBD_DT_COOP_G"] = "8"                  # eight rows per wave (windows of eight entries)
        elif coop is not None:
            env["PBD_DT_COOP"] = coop
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        results[shift + ("" if coop is None else "/coop" + coop)] = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(v == results["0"] for v in results.values()), results
    assert results["0"]["u8"][0] > 10 and results["0"]["i16"][0] > 10
    # and the 64-rows-per-wave result is the oracle's
    for name, shape in (("u8", (150, 330)), ("i16", (60, 1100))):
        model = M.synthetic_model(seed=31, pa=[0, 1, 1, 2, 2, 3], nmix=3, sbin=4, interval=4, thresh=-0.35, linear_def=(name == "i16"), name="narrow")
        want = oracle.detect(model.flatten(), synth.synthetic_frame(47, shape[0], shape[1], 3))
        h = hashlib.sha256()
        for w in want:
            h.update(np.asarray([w["level"], w["component"], w["root_x"], w["root_y"]], np.int32).tobytes()); h.update(np.float32(w["score"]).tobytes())
            h.update(np.ascontiguousarray(w["parts"], dtype=np.int32).tobytes())
        assert [len(want), h.hexdigest()] == results["0"][name], (name, len(want), results["0"][name])
