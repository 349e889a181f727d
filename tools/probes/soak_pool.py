import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from partsbaseddetector_amd import detector, synth, model as M
model = M.synthetic_person_model()
fr = np.stack([synth.synthetic_frame(i + 1, 240, 320, 3) for i in range(4)])
d = torch.from_numpy(fr).cuda()
free0 = None
ref = None
for it in range(40):
    pool = detector.DetectorPool(model, n=3, device=0, max_batch=4, max_candidates=1 << 16)
    for k in range(6):
        while pool.ready_before_next_submit:
            buf, n = pool.wait_batch(raw=True)
        pool.submit_batch_device(d.data_ptr(), 4, 240, 320, 3)
    while pool.pending:
        buf, n = pool.wait_batch(raw=True)
    sig = (n, int(np.asarray(buf[: n * pool.dets[0].hd.stride]).astype(np.int64).sum()))
    if ref is None: ref = sig
    assert sig == ref, (it, sig, ref)
    pool.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it % 10 == 9: print(it, "free GB", round(free / 2**30, 2), flush=True)
assert free0 - free < (1 << 30), ("leak?", free0, free)
print("soak ok", ref)
