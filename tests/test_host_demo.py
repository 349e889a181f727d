"""C++ host mirror (include/pbd_host.hpp) and the demo harness (host/demo.cpp, flow of the reference's
src/demo.cpp): model file -> deserialize -> distributeModel -> detect -> sort [-> NMS]."""
import os
import subprocess

import numpy as np
import pytest

from partsbaseddetector_amd import filestorage as FS
from partsbaseddetector_amd import model as M, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "host", "pbd_demo")


@pytest.fixture(scope="module")
def demo():
    from partsbaseddetector_amd import build
    build.build_hip()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    return DEMO


def _write_inputs(tmp_path, model, im):
    mpath, ipath = str(tmp_path / "model.yml"), str(tmp_path / "frame.ppm")
    FS.serialize(model, mpath)
    with open(ipath, "wb") as fh:
        if im.shape[2] == 3:
            fh.write(b"P6\n%d %d\n255\n" % (im.shape[1], im.shape[0]))
            fh.write(np.ascontiguousarray(im[:, :, ::-1]).tobytes())      # PPM stores RGB; the demo hands BGR to detect()
        else:
            fh.write(b"P5\n%d %d\n255\n" % (im.shape[1], im.shape[0]))
            fh.write(np.ascontiguousarray(im).tobytes())
    return mpath, ipath


def _parse(out):
    lines = out.strip().splitlines()
    n = int([ln for ln in lines if ln.startswith("Number of candidates")][0].split(":")[1])
    cands = []
    for ln in lines:
        if not ln.startswith("cand "):
            continue
        t = ln.split()
        cands.append(((int(t[1]), int(t[2]), int(t[3]), int(t[4])), np.float32(t[5]),
                      np.array([[int(v) for v in p.split(",")] for p in t[6:]], np.int32)))
    return n, cands


def test_demo_fails_loudly_without_gpu(demo, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    mpath, ipath = _write_inputs(tmp_path, M.synthetic_tiny_model(thresh=0.7), synth.synthetic_frame(1, 96, 80, 3))
    r = subprocess.run([demo, mpath, ipath], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU path" in r.stderr       # the model file parsed; pbd_create refused
    r = subprocess.run([demo, mpath, str(tmp_path / "missing.ppm")], capture_output=True, text=True)
    assert r.returncode != 0 and "Image not found" in r.stderr


def _dump_digest(model):
    """what `pbd_demo --dump-model` prints for `model` (every field FileStorageModel::deserialize fills)"""
    out = [f"name {model.name}", f"interval {model.interval}", f"thresh {float(np.float32(model.thresh)):.9g}", f"sbin {model.sbin}",
           f"norient {model.norient}", f"flen {model.flen}"]
    for i, f in enumerate(model.filtersw):
        f = np.asarray(f, np.float64)
        out.append(f"filter {i} {f.shape[0]} {f.shape[1]} " + " ".join("%.17g" % v for v in f.ravel()))
    out.append(("biasw " + " ".join("%.9g" % float(np.float32(b)) for b in model.biasw)).rstrip())
    out.append(("anchors " + " ".join(f"{a[0]},{a[1]}" for a in model.anchors)).rstrip())
    for d, w in enumerate(model.defw):
        out.append(f"def {d} " + " ".join("%.9g" % float(np.float32(v)) for v in w))
    for c in range(model.ncomponents()):
        for p in range(model.nparts(c)):
            line = f"part {c} {p} parent {model.parentid[c][p]} filterid " + " ".join(str(v) for v in model.filterid[c][p])
            line += " biasid" + "".join(f" {v}" for v in model.biasid[c][p]) + " defid" + "".join(f" {v}" for v in model.defid[c][p])
            out.append(line)
    return out


@pytest.mark.parametrize("which", ["tiny", "face", "person"])
def test_cpp_reader_yaml_and_xml(demo, tmp_path, which):
    """pbdhost::FileStorageModel::deserialize (src/FileStorageModel.cpp:96-159) on the YAML and on the XML flavour of
    the same model (the reference's configs name XML models: conf/config_person.by_parts:30): both must hand back
    every field exactly -- checked through `pbd_demo <model> --dump-model`, which needs no GPU."""
    model = {"tiny": M.synthetic_tiny_model(thresh=-0.25), "face": M.synthetic_face_model(thresh=1.5, nparts=7, ncomponents=2),
             "person": M.synthetic_person_model()}[which]
    want = _dump_digest(model)
    for ext, writer in (("yml", FS.serialize), ("xml", FS.serialize_xml)):
        path = str(tmp_path / f"model.{ext}")
        writer(model, path)
        r = subprocess.run([demo, path, "--dump-model"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = [ln.rstrip() for ln in r.stdout.strip().splitlines()]
        assert len(got) == len(want), (ext, len(got), len(want))
        for a, b in zip(got, want):
            assert a.split() == b.split(), (ext, a[:120], b[:120])


def test_cpp_reader_rejects_malformed_xml(demo, tmp_path):
    bad = tmp_path / "bad.xml"
    bad.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<interval>5</interval>\n<thresh>0.</thresh>')   # unterminated
    r = subprocess.run([demo, str(bad), "--dump-model"], capture_output=True, text=True)
    assert r.returncode != 0 and "model file" in r.stderr
    # a closing tag cut off before its '>' (used to restart the scan at offset 0 and recurse until the stack ran out)
    bad.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<interval>5</interval>\n<thresh>0.</thresh>\n</opencv_storage')
    r = subprocess.run([demo, str(bad), "--dump-model"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 254 and "unterminated closing tag" in r.stderr          # -2: a clean pbdhost::Error, not a crash
    # a closing tag that does not match its opening tag
    bad.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<interval>5</thresh>\n</opencv_storage>')
    r = subprocess.run([demo, str(bad), "--dump-model"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 254 and "closed by" in r.stderr
    # a processing instruction inside an element is skipped, not parsed as a child
    ok = tmp_path / "pi.xml"
    FS.serialize_xml(M.synthetic_tiny_model(thresh=0.5), str(ok))
    text = ok.read_text().replace("<interval>", "<?note inside?>\n<interval>", 1)
    ok.write_text(text)
    r = subprocess.run([demo, str(ok), "--dump-model"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "interval" in r.stdout, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flags,dtype", [([], np.float32), (["--staged"], np.float32), (["--double"], np.float64),
                                         (["--double", "--staged"], np.float64),
                                         (["--stream", "3", "10"], np.float32), (["--double", "--stream", "2", "5"], np.float64)])
def test_demo_matches_oracle(demo, oracle, tmp_path, flags, dtype):
    """--stream K N: the frame N times through pbdhost::FrameStream (K handles fed round-robin); the demo itself checks that all
    N results are identical and prints the first, which must be the oracle's like a plain detect()."""
    model = M.synthetic_tiny_model(thresh=0.7)
    im = synth.synthetic_frame(5, 96, 128, 3)
    mpath, ipath = _write_inputs(tmp_path, model, im)
    r = subprocess.run([demo, mpath, ipath] + flags, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if "--stream" in flags:
        assert "results identical" in r.stdout
    n, got = _parse(r.stdout)
    want = oracle.detect(model.flatten(), im, dtype=dtype)
    assert n == len(want) == len(got)
    want_map = {(w["level"], w["component"], w["root_y"], w["root_x"]): w for w in want}
    scores = [g[1] for g in got]
    assert all(a >= b for a, b in zip(scores, scores[1:]))            # Candidate::sort: descending
    for key, score, parts in got:
        w = want_map[key]
        assert np.float32(w["score"]) == score and np.array_equal(parts, w["parts"])


@pytest.mark.gpu
def test_demo_nms_matches_python_mirror(demo, tmp_path):
    from partsbaseddetector_amd import detector as D
    model = M.synthetic_person_model(thresh=17.9)
    im = synth.synthetic_frame(21, 160, 120, 3)
    mpath, ipath = _write_inputs(tmp_path, model, im)
    r = subprocess.run([demo, mpath, ipath, "--nms", "0.1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, got = _parse(r.stdout)
    det = D.PartsBasedDetector(device=0)
    det.distributeModel(model)
    cands = det.detect(im)
    D.Candidate.sort(cands)
    D.Candidate.nonMaximaSuppression(im.shape, cands, 0.1)
    assert len(cands) == len(got) and 0 < len(got)
    assert sorted((c.level, c.root[1], c.root[0]) for c in cands) == sorted((k[0], k[2], k[3]) for k, _, _ in got)
    det.hd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [0.0, 0.1, 0.5])
def test_demo_nms_against_a_painted_pixel_set(demo, tmp_path, overlap):
    """Candidate::sort + nonMaximaSuppression of the C++ host (include/Candidate.hpp:91-111,277-304) on real detections,
    against an INDEPENDENT formulation (not the Python mirror): the canvas is a set of painted (x, y) pixels, a box is the
    set of image pixels inside the hull of the candidate's non-empty part rectangles."""
    model = M.synthetic_person_model(thresh=17.9)
    im = synth.synthetic_frame(21, 160, 120, 3)
    rows, cols = im.shape[:2]
    mpath, ipath = _write_inputs(tmp_path, model, im)
    r = subprocess.run([demo, mpath, ipath], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, allc = _parse(r.stdout)                                   # sorted by score, descending, no suppression
    r = subprocess.run([demo, mpath, ipath, "--nms", repr(overlap)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, got = _parse(r.stdout)
    painted, want = set(), []
    for key, score, parts in allc:
        live = [tuple(int(v) for v in p) for p in parts if p[2] > 0 and p[3] > 0]
        x1 = min(p[0] for p in live); y1 = min(p[1] for p in live)
        x2 = max(p[0] + p[2] for p in live); y2 = max(p[1] + p[3] for p in live)
        box = {(x, y) for x in range(max(x1, 0), min(x2, cols)) for y in range(max(y1, 0), min(y2, rows))}
        if box and len(box & painted) / len(box) > overlap:
            continue
        painted |= box
        want.append(key)
    assert 0 < len(want) < len(allc) or overlap >= 0.5
    assert [k for k, _, _ in got] == want


@pytest.mark.parametrize("ext", ["yml", "xml"])
def test_readers_on_documents_in_opencv_writer_layout(demo, ext):
    """Hand-written documents in the layout OpenCV's FileStorage writer produces -- flow sequences wrapped over lines,
    `!!opencv-matrix` / `type_id="opencv-matrix"` nodes, `<_>` items, an empty `defid` for the root -- committed under
    tests/golden/ (not produced by this repository's writers; no file written by a real OpenCV ships with the reference,
    so the reader stays "parity unpinned").  Python and C++ readers must agree with the values the files were typed from."""
    path = os.path.join(ROOT, "tests", "golden", f"toy_model_opencv_layout.{ext}")
    want_f = np.load(os.path.join(ROOT, "tests", "golden", "toy_model_opencv_layout_filters.npy"))
    m = FS.deserialize(path)
    assert (m.name, m.interval, m.sbin, m.norient, m.flen) == ("toy_opencv_layout", 4, 8, 18, 32) and m.thresh == -0.75
    assert len(m.filtersw) == 3 and all(np.array_equal(a, b) for a, b in zip(m.filtersw, want_f))
    assert np.array_equal(np.float32(m.biasw), np.float32([0.1, -0.2, 0.3, -0.4, 0.5]))
    assert m.anchors == [(1, -2), (0, 3)] and m.parentid == [[-1, 0]]
    assert m.filterid == [[[0], [1, 2]]] and m.biasid == [[[0], [1, 3]]] and m.defid == [[[], [0, 1]]]
    assert np.array_equal(np.float32(m.defw), np.float32([[0.01, 0, 0.02, 0.001], [0.03, -0.001, 0.01, 0]]))
    r = subprocess.run([demo, path, "--dump-model"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [ln.rstrip() for ln in r.stdout.strip().splitlines()]
    want = _dump_digest(m)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.split() == b.split(), (a[:100], b[:100])
