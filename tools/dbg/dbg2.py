import sys
sys.path.insert(0,'oracle'); sys.path.insert(0,'video-steganography-pcamv_amd'); sys.path.insert(0,'tests')
import numpy as np, orc, pcamv_amd, helpers
g=helpers.load("qcif_hex_subme5")
W,H=176,144
def run(t, fresh=True, use_prev=True):
    p=pcamv_amd.param_default(W,H); p.i_mv_range=int(g["mv_range"]); p.i_tscale = 256 if use_prev else 0
    enc=pcamv_amd.Encoder(p)
    prev=(g[f"f{t}_prev_mv"], g[f"f{t}_prev_ref"]) if (f"f{t}_prev_mv" in g and use_prev) else (None,None)
    enc.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev); enc.upload_fenc(g[f"f{t}_fenc_y"], g[f"f{t}_fenc_u"], g[f"f{t}_fenc_v"])
    mbs,rec=enc.analyse_pframe(26,1)
    o=orc.Oracle(orc.make_params(W,H,me="hex",subme=5,mv_range=int(g["mv_range"]),tscale=256 if use_prev else 0))
    o.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev); o.set_fenc(g[f"f{t}_fenc_y"], g[f"f{t}_fenc_u"], g[f"f{t}_fenc_v"])
    mo,ro=o.analyse_pframe(26,1)
    bad=[i for i in range(len(mbs)) if any(not np.array_equal(mbs[f][i],mo[f][i]) for f in mbs.dtype.names)]
    print("frame",t,"use_prev",use_prev,"bad MBs",bad[:20])
    for i in bad[:3]:
        print(" gpu",mbs[i]); print(" orc",mo[i])
    enc.close(); o.close()
run(1); run(2,use_prev=True); run(2,use_prev=False)
