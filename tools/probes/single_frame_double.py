"""One 640x480 frame at a time through PartsBasedDetector<T>::detect, T = float and T = double (the ECTO / ROS callers'
instantiation): milliseconds per frame, host image in, candidates out (synchronous detect(), no pipelining)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from partsbaseddetector_amd import synth
from partsbaseddetector_amd.detector import PartsBasedDetector
from partsbaseddetector_amd.model import synthetic_person_model
model = synthetic_person_model()
im = synth.synthetic_frame(1, 480, 640, 3)
for dt in (np.float32, np.float64):
    det = PartsBasedDetector(device=0, dtype=dt)
    det.distributeModel(model)
    for _ in range(3):
        c = det.detect(im)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        c = det.detect(im)
    ms = (time.perf_counter() - t0) / n * 1e3
    det.hd.profile(1)
    det.detect(im)
    prof = det.hd.profile_read()
    det.hd.profile(0)
    print(np.dtype(dt).name, f"{ms:.3f} ms per frame, {len(c)} candidates;", {k: round(v[0], 3) for k, v in prof.items() if v[1]}, flush=True)
    det.hd.close()
