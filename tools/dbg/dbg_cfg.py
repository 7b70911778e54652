"""tools/dbg/dbg_cfg.py: one RD sweep-style configuration, GPU against the oracle: first differing macroblocks per record field (run on the GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("tests", "oracle", "video-steganography-pcamv_amd"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np, orc, pcamv_amd as pc
from pcamv_amd.synth import make_clip
W, H, me, subme, qp, seed, static, cabac, psy, noise, embed, inter = eval(sys.argv[1])
clip = make_clip(W, H, 3, seed=seed, static_cols=static, noise=noise)
mvr = pc.level_mv_range(W, H)
op = orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter | 1, cabac=cabac, psy_rd=psy)
p = pc.param_default(W, H); pc.param_parse(p, "subme", subme)
p.i_me_method, p.inter, p.i_mv_range, p.b_cabac, p.i_psy_rd, p.i_chroma_qp_offset = pc.ME_NAMES[me], inter | 1, mvr, cabac, op.i_psy_rd, op.i_chroma_qp_offset
enc = pc.Encoder(p); o = orc.Oracle(op); ho = o.debug_state_hash()
if cabac: enc.debug_state_hash(True)
enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1]); o.set_ref(*clip[0]); o.set_fenc(*clip[1])
mbs, rec = enc.analyse_pframe(qp, embed=embed); mo, ro = o.analyse_pframe(qp, embed)
if cabac: print("hash bad", np.nonzero(enc.state_hash_fetch() != ho)[0][:6])
for f in mbs.dtype.names:
    if not np.array_equal(mbs[f], mo[f]):
        bad = np.argwhere((mbs[f] != mo[f]).reshape(len(mbs), -1).any(1)).ravel()
        print(f, bad[:6], "gpu", mbs[f][bad[0]].tolist(), "orc", mo[f][bad[0]].tolist())
print("recon equal", [bool(np.array_equal(a, b)) for a, b in zip(rec, ro)])
if len(sys.argv) > 2:           # trace build (PCAMV_GPU_LIB=..._tr.so): the traced macroblock's evaluations; rows with ip = -1 are the sub-partition RD trials
    mb = int(sys.argv[2])
    enc2 = pc.Encoder(p)
    enc2.set_ref(*clip[0]); enc2.upload_fenc(*clip[1])
    enc2.trace_mb(mb)
    enc2.analyse_pframe(qp, embed=embed)
    tr = enc2.trace_fetch()
    for r in tr:
        if r[0] == -2: print("gpu trial type", r[1], "part", r[2], "cost", r[3], "cbp", r[4], "sub", [(r[5] >> s) & 255 for s in (0, 8, 16, 24)], "counts", r[6])
        if r[0] == -3: print("gpu p_rd thresh", r[1], "cost8x8", r[2], "16x8", r[3], "8x16", r[4], "me16", r[5], "rd16", r[6], "flags", r[7])
        if r[0] == -5: print("gpu p8x8 branch psub_on", r[1], "inter", hex(r[2]), "variant", r[3], "costs[2]", r[4:8].tolist())
        if r[0] == -6: print("gpu ELSE branch psub_on", r[1], "inter", hex(r[2]), "variant", r[3])
        if r[0] == -4: print("gpu sub costs i8", r[1], "4x4", r[2], "8x4", r[3], "4x8", r[4], "8x8", r[5], "th", r[6])
        if r[0] == -1: print("gpu part mb", mb, "i8", r[1], "sub", r[2], "ssd", r[3], "bits", r[4], "cbp", r[5], "mv", r[6], r[7])
    print("trace rows", len(tr), "lists by ip", np.bincount(tr[tr[:, 0] >= 0][:, 0], minlength=7).tolist())
