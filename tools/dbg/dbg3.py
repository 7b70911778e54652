import sys
sys.path.insert(0,'oracle'); sys.path.insert(0,'video-steganography-pcamv_amd'); sys.path.insert(0,'tests')
import numpy as np, orc, pcamv_amd, helpers
from emu import emu
g=helpers.load("qcif_hex_subme5")
W,H=176,144; t=2; MB=int(sys.argv[1]) if len(sys.argv)>1 else 3
p=pcamv_amd.param_default(W,H); p.i_mv_range=int(g["mv_range"])
enc=pcamv_amd.Encoder(p)
prev=(g[f"f{t}_prev_mv"], g[f"f{t}_prev_ref"])
enc.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev); enc.upload_fenc(g[f"f{t}_fenc_y"], g[f"f{t}_fenc_u"], g[f"f{t}_fenc_v"])
enc.trace_mb(MB)
mbs,rec=enc.analyse_pframe(26,0)
tg=enc.trace_fetch()
op=orc.make_params(W,H,me="hex",subme=5,mv_range=int(g["mv_range"]))
o=orc.Oracle(op); o.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev)
me,re_,te=emu.analyse_pframe(orc, op, 26, 0, [g[f"f{t}_fenc_{c}"] for c in "yuv"], o.ref_planes(), g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev, diag=1, trace_mb=MB)
print("gpu evals",len(tg),"emu evals",len(te))
n=min(len(tg),len(te))
for i in range(n):
    if not np.array_equal(tg[i],te[i]):
        print("first diff at",i); print(" gpu",tg[max(0,i-3):i+3].tolist()); print(" emu",te[max(0,i-3):i+3].tolist()); break
else: print("common prefix identical")
