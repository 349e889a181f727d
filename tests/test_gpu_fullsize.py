"""BASELINE.json configs[2] / [4] at their full sizes against the CPU oracle.

configs[2]: person model, a batch of 64 640x480 frames, the whole path on one GPU -- through both batch entry
points of the C ABI (pbd_detect_batch_device: frames resident in HBM; pbd_detect_batch: host pointers), every
candidate of 16 of the frames compared with oracle.detect (first, last and a stride through the middle, chosen so
that both ends of every dynamic-program chunk are hit), and the staged responses / root scores of frame 63,
level 0 compared bit for bit.  Mirrors PartsBasedDetector<T>::detect (reference src/PartsBasedDetector.cpp:69-95)
frame by frame: the reference has no batch API, so the bar is "identical to 64 single calls".
"""
import ctypes as C
import os

import numpy as np
import pytest

from partsbaseddetector_amd import synth
from partsbaseddetector_amd import model as M

pytestmark = pytest.mark.gpu

B, ROWS, COLS = 64, 480, 640


def _key(w):
    return (w["level"], w["component"], w["root_y"], w["root_x"])


def _check_frame(cands, want, what):
    assert len(cands) == len(want), (what, len(cands), len(want))
    for g, w in zip(cands, want):
        assert (g.level, g.component, g.root[1], g.root[0]) == _key(w), what
        assert np.array_equal(g.parts, w["parts"]), what
        assert np.float32(g.score()) == np.float32(w["score"]), what


@pytest.fixture(scope="module")
def vga_batch():
    return np.stack([synth.synthetic_frame(i + 1, ROWS, COLS, 3) for i in range(B)])      # the bench's frames (rank 0)


def test_config2_batch_of_64_vga_frames(oracle, vga_batch, monkeypatch):
    import torch
    from partsbaseddetector_amd import detector, _lib
    model = M.synthetic_person_model()
    flat = model.flatten()
    check = sorted({0, 1, 15, 16, 31, 32, 47, 48, 62, 63} | set(range(5, 64, 9)))      # 16+ frames, chunk ends included
    assert len(check) >= 16
    want = {i: oracle.detect(flat, vga_batch[i]) for i in check}
    assert sum(len(w) for w in want.values()) > 100

    def run(budget_mb):
        # budget_mb forces the dynamic program into chunks of 16 frames (None: the default budget, one chunk)
        if budget_mb is None:
            monkeypatch.delenv("PBD_DP_BUDGET_MB", raising=False)
        else:
            monkeypatch.setenv("PBD_DP_BUDGET_MB", str(budget_mb))
        det = detector.PartsBasedDetector(device=0, max_batch=B, max_candidates=1 << 16)
        det.distributeModel(model)
        d_frames = torch.from_numpy(vga_batch).cuda()
        dev = det.detect_batch_device(d_frames.data_ptr(), B, ROWS, COLS, 3)
        for i in check:
            _check_frame([c for c in dev if c.frame == i], want[i], f"device entry, frame {i}, budget {budget_mb}")
        assert all(0 <= c.frame < B for c in dev)
        # staged read-back of the same run: frame 63, level 0
        plan = det.hd.plan(ROWS, COLS)
        H, W = int(plan["feat_rows"][0]), int(plan["feat_cols"][0])
        feats, _ = oracle.features_pyramid(flat, vga_batch[63])
        got_r = det.hd.get_stage(_lib.STAGE_RESPONSES, 63, 0, H, W)
        assert np.array_equal(got_r.view(np.uint32), oracle.responses(flat, feats[0]).view(np.uint32))
        _, _, _, orv, ori = oracle.dp_min(flat, 0, got_r)
        assert np.array_equal(det.hd.get_stage(_lib.STAGE_ROOTV, 63, 0, H, W)[0].view(np.uint32), orv.view(np.uint32))
        assert np.array_equal(det.hd.get_stage(_lib.STAGE_ROOTI, 63, 0, H, W)[0], ori)
        # the host-pointer entry point on the same frames: the identical record list
        host = det.detect_batch([vga_batch[i] for i in range(B)])
        assert len(host) == len(dev)
        for a, b in zip(host, dev):
            assert (a.frame, a.level, a.component, a.root) == (b.frame, b.level, b.component, b.root)
            assert np.array_equal(a.parts, b.parts) and a.score() == b.score()
        det.hd.close()
        del d_frames
        return len(dev)

    n1 = run(None)
    n2 = run(900)          # ~47 MB of scratch per frame -> chunks of 16 frames
    assert n1 == n2


def test_f64_batch_with_several_dp_chunks(oracle, monkeypatch):
    """PBD_REAL_F64 with a budget that forces dp_chunk_frames < nframes: 6 frames 240x320, chunks of 2."""
    from partsbaseddetector_amd import detector
    monkeypatch.setenv("PBD_DP_BUDGET_MB", "60")
    model = M.synthetic_person_model(thresh=18.0)
    flat = model.flatten()
    frames = [synth.synthetic_frame(40 + i, 240, 320, 3) for i in range(6)]
    det = detector.PartsBasedDetector(device=0, max_batch=6, dtype=np.float64)
    det.distributeModel(model)
    got = det.detect_batch(frames)
    total = 0
    for i, f in enumerate(frames):
        want = oracle.detect(flat, f, dtype=np.float64)
        total += len(want)
        _check_frame([c for c in got if c.frame == i], want, f"f64 frame {i}")
    assert total > 0 and total == len(got)
    det.hd.close()


def test_config4_mfma_f16_one_vga_frame(oracle, vga_batch):
    """BASELINE configs[4] / SURVEY 8(d) config 5 at VGA: fp16 operands on the matrix cores.  The 1e-4 bar does not
    apply; reported instead: max-abs response error vs the reference-order fp32 convolution and the arg-max
    (root set, root mixture, part placement) agreement with the exact path."""
    from partsbaseddetector_amd import detector, _lib
    model = M.synthetic_person_model()
    flat = model.flatten()
    im = vga_batch[0]
    res = {}
    worst = 0.0
    for name, mode in (("exact", _lib.CONV_EXACT), ("f16", _lib.CONV_MFMA_F16)):
        det = detector.PartsBasedDetector(device=0, conv_mode=mode)
        det.distributeModel(model)
        res[name] = {(c.level, c.component, c.root[1], c.root[0]): c for c in det.detect(im)}
        if name == "f16":
            feats, _ = oracle.features_pyramid(flat, im)
            for l in (0, 10, 25, len(feats) - 1):
                H, W = feats[l].shape[0], feats[l].shape[1] // 32
                r = det.hd.get_stage(_lib.STAGE_RESPONSES, 0, l, H, W)
                ref = oracle.responses(flat, feats[l])
                worst = max(worst, float(np.abs(r - ref).max()))
                assert np.all(np.abs(r - ref) <= 2e-3 + 2.0 ** -10 * np.abs(ref))       # fp16 operands + fp16 responses
        det.hd.close()
    common = set(res["exact"]) & set(res["f16"])
    same = sum(np.array_equal(res["exact"][k].parts, res["f16"][k].parts) for k in common)
    dscore = max((abs(res["exact"][k].score() - res["f16"][k].score()) for k in common), default=0.0)
    print(f"configs[4] VGA: max |response - reference| = {worst:.3g}; roots exact {len(res['exact'])}, fp16 {len(res['f16'])}, "
          f"common {len(common)}, identical part placements {same}, max score diff {dscore:.3g}")
    assert len(res["exact"]) > 0
    assert worst <= 1e-2
    assert len(common) >= 0.9 * max(len(res["exact"]), len(res["f16"]))
    assert same >= 0.5 * len(common)      # part placements move on near ties at fp16 precision: the rate is the reported figure
