import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import oracle
from partsbaseddetector_amd import model as M, synth
oracle.build()
model=M.synthetic_person_model(); flat=model.flatten()
im=synth.synthetic_frame(1,480,640,3)
feats=oracle.features_pyramid(flat, im)
lv=0
resp=oracle.responses(flat, feats[lv] if isinstance(feats,list) else feats[0][lv])
print(resp.shape)
a=-0.01
def sim(row, T):
    # returns (#elements, spilled entries, reloaded entries, forward pops, readout pops) for ring depth T (pairs spill)
    N=len(row); v=[0]; z=[-np.inf]; lo=0; spills=0; reloads=0; fp=0
    for q in range(1,N):
        while True:
            k=len(v)-1
            s=((row[q]-row[v[k]])+a*(q*q-v[k]*v[k]))/(2*a*(q-v[k]))
            if s<=z[k] and k>0:
                v.pop(); z.pop(); fp+=1
                if len(v)-1 < lo and lo>0:   # top index below ring start: reload pair
                    lo-=2; reloads+=2
            else: break
        v.append(q); z.append(s)
        if (len(v)-1) - lo > T:    # entries below top exceed ring
            lo+=2; spills+=2
    # readout pops everything down
    rp=len(v)-1
    # reloads during readout: all spilled entries still there
    reloads+=lo
    return N, spills, reloads, fp, rp
rng=np.random.default_rng(0)
tot={T:[0,0,0] for T in (4,8,12,16,24,32)}
fpt=0; rpt=0; n=0
for f in rng.choice(resp.shape[0], 6, replace=False):
    for y in range(0, resp.shape[1], 7):
        row=resp[f,y].astype(np.float64)
        for T in tot:
            N,sp,rl,fp,rp=sim(row,T)
            tot[T][0]+=N; tot[T][1]+=sp; tot[T][2]+=rl
        fpt+=fp; rpt+=rp; n+=N
for T,(N,sp,rl) in tot.items():
    print('ring',T,'spilled entries/element %.3f'%(sp/N),'reloaded/element %.3f'%(rl/N))
print('forward pops/element %.3f readout pops %.3f'%(fpt/n, rpt/n))
