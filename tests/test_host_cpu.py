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
