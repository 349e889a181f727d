"""One-off wider run of tests/test_gpu_fuzz.py's generator (seeds beyond the committed 20); run on the GPU box:
python tools/fuzz_more.py <first> <last>"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from oracle import oracle                       # noqa: E402  (checker)
from partsbaseddetector_amd import detector     # noqa: E402

oracle.build()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    model, im = fz._random_case(seed)
    flat = model.flatten()
    try:
        want = oracle.detect(flat, im, capacity=1500000)
    except RuntimeError:
        # frame too small for one octave (the reference would index out of bounds): the library must refuse it too
        det = detector.PartsBasedDetector(device=0)
        det.distributeModel(model)
        try:
            det.detect(im)
            print("NOT REJECTED seed", seed, im.shape, flush=True)
            bad += 1
        except detector.PbdError:
            pass
        det.hd.close()
        continue
    if len(want) > 200:
        model.thresh = float(np.sort(np.array([w["score"] for w in want], np.float32))[-200])
        flat = model.flatten()
        want = oracle.detect(flat, im)
    for dtype in (np.float32, np.float64) if seed % 5 == 0 else (np.float32,):
        w2 = want if dtype == np.float32 else oracle.detect(flat, im, dtype=np.float64)
        det = detector.PartsBasedDetector(device=0, dtype=dtype)
        det.distributeModel(model)
        got = det.detect(im)
        det.hd.close()
        ok = len(got) == len(w2) and all(
            (g.level, g.component, g.root[1], g.root[0]) == (w["level"], w["component"], w["root_y"], w["root_x"])
            and np.array_equal(g.parts, w["parts"]) and dtype(g.score()) == dtype(w["score"]) for g, w in zip(got, w2))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, dtype.__name__, im.shape, flat.nfilters, len(got), len(w2), flush=True)
print("seeds", first, last, "mismatches", bad)
