"""Single-frame stream over K handles (K streams, K workspaces) on one GPU: frames alternate between the handles, each handle one
frame deep.  Prints frames per second for K = 1, 2, 3, 4.   python tools/probes/two_handles.py [rows cols batch]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from partsbaseddetector_amd import synth, _lib
from partsbaseddetector_amd.detector import PartsBasedDetector
from partsbaseddetector_amd.model import synthetic_person_model

rows, cols, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640, 1)
model = synthetic_person_model()
frames = np.stack([synth.synthetic_frame(i + 1, rows, cols, 3) for i in range(B)])
d = torch.from_numpy(frames).cuda()
for K in (1, 2, 3, 4):
    dets = []
    for k in range(K):
        det = PartsBasedDetector(device=0, max_batch=B, max_candidates=1 << 16)
        det.distributeModel(model)
        dets.append(det)
    def run(n):
        inflight = [False] * K
        got = 0
        for i in range(n):
            k = i % K
            if inflight[k]:
                _, nc = dets[k].wait_batch(raw=True); got += nc
            dets[k].submit_batch_device(d.data_ptr(), B, rows, cols, 3)
            inflight[k] = True
        for k in range(K):
            if inflight[k]:
                _, nc = dets[k].wait_batch(raw=True); got += nc
        return got
    run(8)
    torch.cuda.synchronize()
    n = 200 if B == 1 else 40
    t0 = time.perf_counter(); got = run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"handles {K}: {n * B / dt:8.1f} frames/s  {dt / n * 1e3:7.3f} ms per submit   candidates {got}", flush=True)
    for det in dets:
        det.hd.close()
