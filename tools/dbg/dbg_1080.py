import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "tests"), os.path.join(R, "oracle"), os.path.join(R, "video-steganography-pcamv_amd")]
import orc, helpers, pcamv_amd as pc
from pcamv_amd.synth import make_clip
W, H = 1920, 1088
clip = make_clip(W, H, 3, seed=13)
mvr = pc.level_mv_range(W, H)
p = pc.param_default(W, H); p.i_me_method = pc.ME_NAMES["umh"]; p.i_subpel_refine = 5; p.inter = 0x10; p.i_mv_range = mvr; p.i_tscale = 256
enc = pc.Encoder(p)
enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1])
mbs, rec = enc.analyse_pframe(26, embed=1)
emb = enc.embed_pframe(0.5)
o = orc.Oracle(orc.make_params(W, H, me="umh", subme=5, mv_range=mvr))
o.set_ref(*clip[0]); o.set_fenc(*clip[1])
mbs_o, _ = o.analyse_pframe(26, 1)
emb_o = o.embed_pframe(mbs_o, 0.5)
print("hdr", emb["n"], emb["m"], emb["stc_ok"], emb["num_flip"], "|", emb_o["n"], emb_o["m"], emb_o["stc_ok"], emb_o["num_flip"])
for k in ("cover", "rho", "message", "stego", "flip"):
    a, b = np.asarray(emb[k]), np.asarray(emb_o[k])
    d = np.argwhere(a != b).ravel()
    print(k, len(a), len(b), "ndiff", len(d), d[:10], a[d[:5]], b[d[:5]])
ext = pc.stc_extract(helpers.carrier_lsbs(enc.final_mvs(mbs)), emb["m"])
print("gpu BER", float((ext != emb["message"]).mean()))
import ctypes
st_o = np.asarray(emb_o["stego"]); ext_o = pc.stc_extract(st_o, emb_o["m"])
print("oracle stego extract BER", float((ext_o != emb_o["message"]).mean()))
rho = np.asarray(emb["rho"], np.float64); cov = np.asarray(emb["cover"])
print("distortion gpu", rho[np.asarray(emb["stego"]) != cov].sum(), "oracle", rho[st_o != cov].sum())
