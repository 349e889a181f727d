"""Agreement of the matrix-core convolution mode with the exact mode on the bench workload:
responses (max abs difference) and the final candidates (identical records?)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from partsbaseddetector_amd import _lib, synth
from partsbaseddetector_amd.detector import PartsBasedDetector
from partsbaseddetector_amd.model import synthetic_person_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synthetic_person_model()
frames = np.stack([synth.synthetic_frame(i + 1, 480, 640, 3) for i in range(B)])
d = torch.from_numpy(frames).cuda()
out = {}
for name, mode in (("exact", _lib.CONV_EXACT), ("mfma", _lib.CONV_MFMA)):
    det = PartsBasedDetector(device=0, conv_mode=mode, max_batch=B, max_candidates=1 << 16)
    det.distributeModel(model)
    buf, n = det.detect_batch_device(d.data_ptr(), B, 480, 640, 3, raw=True)
    rec = buf[:n * det.hd.stride].reshape(n, det.hd.stride).copy()
    plan = det.hd.plan(480, 640)
    resp = [det.hd.get_stage(1, 0, l, int(plan["feat_rows"][l]), int(plan["feat_cols"][l])) for l in (0, 10, 25)]
    rootv = [det.hd.get_stage(2, 0, l, int(plan["feat_rows"][l]), int(plan["feat_cols"][l])) for l in (0, 10, 25)]
    out[name] = (rec, resp, rootv)
    det.hd.close()
a, b = out["exact"], out["mfma"]
print("candidates exact/mfma:", len(a[0]), len(b[0]))
ka = {tuple(r[[0, 1, 2, 3, 4]]): r for r in a[0]}
kb = {tuple(r[[0, 1, 2, 3, 4]]): r for r in b[0]}
common = set(ka) & set(kb)
same_boxes = sum(np.array_equal(ka[k][8:], kb[k][8:]) for k in common)
print("common roots:", len(common), "identical part boxes:", same_boxes, "only exact:", len(ka) - len(common), "only mfma:", len(kb) - len(common))
print("max |response diff|:", max(float(np.abs(x - y).max()) for x, y in zip(a[1], b[1])))
print("max |rootv diff|:", max(float(np.abs(x - y).max()) for x, y in zip(a[2], b[2])))
sd = [abs(float(ka[k][5:6].view(np.float32)[0]) - float(kb[k][5:6].view(np.float32)[0])) for k in common]
print("max |score diff| over common candidates:", max(sd) if sd else None)
