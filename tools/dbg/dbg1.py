import sys
sys.path.insert(0,'oracle'); sys.path.insert(0,'video-steganography-pcamv_amd')
import numpy as np, orc, pcamv_amd
from pcamv_amd.synth import make_clip
W,H=176,144
clip=make_clip(W,H,2,seed=5)
o=orc.Oracle(orc.make_params(W,H,mv_range=64)); o.set_ref(*clip[0]); P=o.ref_planes()
p=pcamv_amd.param_default(W,H); enc=pcamv_amd.Encoder(p); enc.set_ref(*clip[0]); G=enc.ref_planes()
for k in range(4):
    d=np.argwhere(G[k]!=P[k]); print("plane",k,len(d), d[:8].tolist(), d[-3:].tolist())
    if len(d):
        y,x=d[0]; print(" got",G[k][y,x:x+8].tolist()," exp",P[k][y,x:x+8].tolist())
        ys=np.unique(d[:,0]); xs=np.unique(d[:,1]); print(" rows",ys[:10],ys[-5:],"cols",xs[:10],xs[-5:])
