import sys, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
from partsbaseddetector_amd import detector as D, model as M, synth, _lib
model = M.synthetic_person_model(thresh=18.9)
hd = D.Handle(model, device=0, real_type=_lib.REAL_F64)
im = synth.synthetic_frame(21, 480, 640, 3)
f = D.HOGFeatures(hd); feats = f.pyramid(im); print("features ok", len(feats), flush=True)
c = D.SpatialConvolutionEngine(hd); resp = c.pdf(feats); print("conv ok", resp[0].shape, flush=True)
dp = D.DynamicProgram(hd); out = dp.min(resp); print("dp ok", flush=True)
cands = dp.argmin(f.scales()); print("argmin ok", len(cands), flush=True)
