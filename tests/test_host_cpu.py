"""Host-side pieces around the path (no GPU): model file format (src/FileStorageModel.cpp) and the
callers' post-step Candidate::sort + nonMaximaSuppression (include/Candidate.hpp:91-111,277-304)."""
import numpy as np
import pytest

from partsbaseddetector_amd import filestorage as FS
from partsbaseddetector_amd import model as M


def _same_model(a, b):
    assert (a.name, a.interval, a.sbin, a.norient, a.flen) == (b.name, b.interval, b.sbin, b.norient, b.flen)
    assert np.float32(a.thresh) == np.float32(b.thresh)
    assert len(a.filtersw) == len(b.filtersw)
    for x, y in zip(a.filtersw, b.filtersw):
        assert np.array_equal(np.asarray(x, np.float64), np.asarray(y, np.float64))     # doubles survive the text round trip
    assert np.array_equal(np.float32(a.biasw), np.float32(b.biasw))
    assert [tuple(v) for v in a.anchors] == [tuple(v) for v in b.anchors]
    assert np.array_equal(np.float32(a.defw), np.float32(b.defw))
    assert a.filterid == b.filterid and a.parentid == b.parentid and a.defid == b.defid
    for ca, cb in zip(a.biasid, b.biasid):
        for pa, pb in zip(ca, cb):
            assert list(pa) == list(pb)


@pytest.mark.parametrize("which", ["tiny", "face", "person"])
def test_yaml_roundtrip(tmp_path, which):
    model = {"tiny": M.synthetic_tiny_model(thresh=-0.25), "face": M.synthetic_face_model(thresh=1.5, nparts=7, ncomponents=2),
             "person": M.synthetic_person_model()}[which]
    path = str(tmp_path / "model.yml")
    assert FS.serialize(model, path)
    back = FS.deserialize(path)
    _same_model(model, back)
    # multi-mixture defid survives (the reference fork's isInt() shortcut would collapse it: Appendix D.3)
    if which != "face":
        assert len(back.defid[0][1]) == len(model.filterid[0][1]) > 1
    # flattening the re-read model gives the same tables
    fa, fb = model.flatten(), back.flatten()
    for name in ("filters_f32", "biasw", "defw", "anchors", "parentid", "filterid", "biasid", "defid", "mix_offset"):
        assert np.array_equal(getattr(fa, name), getattr(fb, name)), name


def test_xml_document(tmp_path):
    """the <opencv_storage> flavour of the same document (what `fs.open("x.xml", WRITE)` produces)"""
    rng = np.random.default_rng(0)
    f0 = rng.standard_normal((5, 160))
    xml = f"""<?xml version="1.0"?>
<opencv_storage>
<name>"toy"</name>
<interval>5</interval>
<thresh>-1.5000000000000000e+00</thresh>
<sbin>4</sbin>
<norient>18</norient>
<flen>32</flen>
<filtersw>
  <_ type_id="opencv-matrix">
    <rows>5</rows>
    <cols>160</cols>
    <dt>d</dt>
    <data>
      {" ".join(repr(float(v)) for v in f0.ravel())}</data></_>
  <_ type_id="opencv-matrix">
    <rows>5</rows>
    <cols>160</cols>
    <dt>d</dt>
    <data>
      {" ".join(repr(float(v)) for v in (2 * f0).ravel())}</data></_></filtersw>
<biasw>
  1.00000001e-01 -2.00000003e-01</biasw>
<anchors>
  1 -2</anchors>
<defs>
  <_>
    9.99999978e-03 0. 9.99999978e-03 0.</_></defs>
<indexers>
  <component-0>
    <part-0>
      <parentid>-1</parentid>
      <filterid>
        0</filterid>
      <biasid>
        0</biasid>
      <defid></defid></part-0>
    <part-1>
      <parentid>0</parentid>
      <filterid>
        1</filterid>
      <biasid>
        1</biasid>
      <defid>
        0</defid></part-1></component-0></indexers>
</opencv_storage>
"""
    path = tmp_path / "toy.xml"
    path.write_text(xml)
    m = FS.deserialize(str(path))
    assert m.name == "toy" and m.interval == 5 and m.thresh == -1.5 and m.nparts(0) == 2
    assert np.array_equal(m.filtersw[0], f0) and np.array_equal(m.filtersw[1], 2 * f0)
    assert m.anchors == [(1, -2)] and m.parentid == [[-1, 0]] and m.defid == [[[], [0]]]
    assert np.allclose(m.defw, [[0.01, 0, 0.01, 0]])
    m.flatten()


def _cand(parts, score):
    from partsbaseddetector_amd.detector import Candidate
    conf = np.zeros(len(parts), np.float32)
    conf[0] = score
    return Candidate(parts=np.asarray(parts, np.int32), confidence=conf, component=0)


def test_sort_and_nms():
    from partsbaseddetector_amd.detector import Candidate
    a = _cand([(10, 10, 20, 20), (25, 12, 10, 10)], 3.0)      # hull (10,10,25,20)
    b = _cand([(12, 12, 20, 20)], 2.0)                         # mostly inside a's hull
    c = _cand([(60, 60, 10, 10)], 2.5)                         # far away
    d = _cand([(-30, -30, 10, 10)], 9.0)                       # entirely outside the image: empty box, kept
    e = _cand([(30, 25, 20, 20)], 1.0)                         # overlaps a's hull by 25 of 400 pixels
    assert a.boundingBox() == (10, 10, 25, 20)
    cands = [b, a, c, d, e]
    Candidate.sort(cands)
    assert [x.score() for x in cands] == [9.0, 3.0, 2.5, 2.0, 1.0]
    kept = list(cands)
    Candidate.nonMaximaSuppression((100, 100), kept, 0.0)
    assert [x.score() for x in kept] == [9.0, 3.0, 2.5]         # b and e touch painted pixels
    kept = list(cands)
    Candidate.nonMaximaSuppression((100, 100), kept, 0.1)       # callers' default (cells/detect.cpp:124, ros/Node.cpp:196)
    assert [x.score() for x in kept] == [9.0, 3.0, 2.5, 1.0]    # e: 25/400 = 0.0625 <= 0.1 kept; b: 0.9 suppressed


def test_nms_against_a_painted_pixel_set():
    """Independent formulation of include/Candidate.hpp:277-304 (not the implementation under test run twice): the
    canvas is a Python set of painted (x, y) pixels; a candidate's box is the set of image pixels inside the hull of
    its non-empty part rectangles.  Random candidates incl. boxes partly / entirely outside the image, zero-area parts
    and duplicates; several overlap thresholds."""
    from partsbaseddetector_amd.detector import Candidate
    rng = np.random.default_rng(7)
    rows, cols = 60, 80
    for trial in range(30):
        cands = []
        for i in range(int(rng.integers(1, 25))):
            nparts = int(rng.integers(1, 5))
            parts = []
            for _ in range(nparts):
                x, y = int(rng.integers(-30, cols + 10)), int(rng.integers(-30, rows + 10))
                w, h = int(rng.integers(0, 40)), int(rng.integers(0, 40))        # zero width / height happens
                parts.append((x, y, w, h))
            if parts[0][2] == 0 or parts[0][3] == 0:
                parts[0] = (parts[0][0], parts[0][1], max(parts[0][2], 1), max(parts[0][3], 1))
            cands.append(_cand(parts, float(rng.standard_normal())))
        if trial % 3 == 0 and cands:
            cands.append(_cand([tuple(int(v) for v in p) for p in cands[0].parts], cands[0].score() - 1e-3))   # a duplicate box
        Candidate.sort(cands)
        assert all(a.score() >= b.score() for a, b in zip(cands, cands[1:]))
        for overlap in (0.0, 0.1, 0.5):
            painted, want = set(), []
            for c in cands:
                px = set()
                live = [(x, y, w, h) for x, y, w, h in (tuple(int(v) for v in p) for p in c.parts) if w > 0 and h > 0]
                x1 = min(p[0] for p in live); y1 = min(p[1] for p in live)
                x2 = max(p[0] + p[2] for p in live); y2 = max(p[1] + p[3] for p in live)
                box = {(x, y) for x in range(max(x1, 0), min(x2, cols)) for y in range(max(y1, 0), min(y2, rows))}
                if box and len(box & painted) / len(box) > overlap:
                    continue                         # suppressed
                painted |= box                       # an empty box (outside the image) is kept and paints nothing
                want.append(id(c))
            kept = list(cands)
            Candidate.nonMaximaSuppression((rows, cols), kept, overlap)
            assert [id(c) for c in kept] == want, (trial, overlap)


def test_by_parts_config(tmp_path):
    """BASELINE configs[0] plumbing: a `.by_parts` pipeline description -> pipeline1.parameters.extra.model_file
    (reference conf/config_face.by_parts, cells/detect.cpp:115-126) -> model."""
    from partsbaseddetector_amd import config as CFG
    model = M.synthetic_face_model(thresh=1.5, nparts=5, ncomponents=2)
    FS.serialize_xml(model, str(tmp_path / "Face_5parts.xml"))
    text = """
source1:
  type: RosKinect
  module: object_recognition_ros.io
sink1:
  type: Publisher
  module: 'object_recognition_by_parts'
pipeline1:
  type: PartsBasedDetector
  module: 'object_recognition_by_parts'
  inputs: [source1]
  outputs: [sink1]
  parameters:
    object_ids: ['abc']
    visualize: true
    extra:
        model_file: "/somewhere/else/models/Face_5parts.xml"
        use_cuda: false
"""
    path = tmp_path / "config_face.by_parts"
    path.write_text(text)
    cfgs = CFG.load_by_parts(str(path))
    assert len(cfgs) == 1
    c = cfgs[0]
    assert c.pipeline == "pipeline1" and c.model_file.endswith("Face_5parts.xml") and c.visualize and not c.use_cuda
    assert c.max_overlap == pytest.approx(0.1)                                   # cells/detect.cpp:124 default
    with pytest.raises(FileNotFoundError):
        CFG.load_model(c)                                                        # the authors' absolute path
    _same_model(model, CFG.load_model(c, search_dirs=[str(tmp_path)]))           # same file name next to the config
    stand_in = M.synthetic_face_model()
    assert CFG.load_model(c, stand_in=stand_in) is stand_in                      # no file: synthetic stand-in
    (tmp_path / "bad.by_parts").write_text("pipeline1:\n  type: PartsBasedDetector\n  parameters: {}\n")
    with pytest.raises(ValueError):
        CFG.load_by_parts(str(tmp_path / "bad.by_parts"))


@pytest.mark.parametrize("which", ["tiny", "face"])
def test_xml_writer_roundtrip(tmp_path, which):
    model = M.synthetic_tiny_model(thresh=-0.25) if which == "tiny" else M.synthetic_face_model(thresh=1.5, nparts=7, ncomponents=2)
    path = str(tmp_path / "model.xml")
    assert FS.serialize_xml(model, path)
    _same_model(model, FS.deserialize(path))


def test_validate_raises_value_error():
    m = M.synthetic_tiny_model()
    m.parentid[0][1] = 2                        # parent after child
    with pytest.raises(ValueError):
        m.validate()
    m = M.synthetic_tiny_model()
    m.filterid[0][2][0] = 99
    with pytest.raises(ValueError):
        m.flatten()
