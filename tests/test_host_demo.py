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
    n = int(lines[0].split(":")[1])
    cands = []
    for ln in lines[1:]:
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


@pytest.mark.gpu
@pytest.mark.parametrize("flags,dtype", [([], np.float32), (["--staged"], np.float32), (["--double"], np.float64),
                                         (["--double", "--staged"], np.float64)])
def test_demo_matches_oracle(demo, oracle, tmp_path, flags, dtype):
    model = M.synthetic_tiny_model(thresh=0.7)
    im = synth.synthetic_frame(5, 96, 128, 3)
    mpath, ipath = _write_inputs(tmp_path, model, im)
    r = subprocess.run([demo, mpath, ipath] + flags, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
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
