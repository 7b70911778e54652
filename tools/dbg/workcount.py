"""Work per macroblock of the bench workload, counted on the CPU emulation of the control code."""
import ctypes as C, sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "tests"), os.path.join(R, "oracle"), os.path.join(R, "video-steganography-pcamv_amd")]
import orc
from emu import emu
from pcamv_amd.synth import make_clip
import pcamv_amd
W, H = 1920, 1088
clip = make_clip(W, H, 3, seed=13)
SUBME = int(sys.argv[1]) if len(sys.argv) > 1 else 7
p = orc.make_params(W, H, me="umh", subme=SUBME, mv_range=pcamv_amd.level_mv_range(W, H), tscale=256, inter=0x11 if SUBME >= 6 else 0x10)
o = orc.Oracle(p)
o.set_ref(*clip[0]); 
lib = C.CDLL(emu.build())
st = (C.c_longlong * 32)()
lib.emu_get_stats(st, 1)
mbs, rec = emu.analyse_pframe(orc, p, 26, 1, list(clip[1]), o.ref_planes(), clip[0][1], clip[0][2], None, None, diag=3 if SUBME >= 6 else 2)
lib.emu_get_stats(st, 1)
n = (W // 16) * (H // 16)
names = ["fpel SAD", "qpel SAD", "SATD", "SATD+chroma"]
for k in range(4):
    print(f"{names[k]:12s} lists/MB {st[k]/n:7.2f}  cands/MB {st[4+k]/n:8.2f}  luma passes/MB {st[8+k]/n:7.2f}")
print(f"chroma passes/MB {st[12]/n:.2f}   lists on recon (RCA) /MB {st[13]/n:.2f}   residual calls/MB {st[14]/n:.2f}")
t = np.bincount(mbs["i_type"], minlength=8); print("types", t, "partitions", np.bincount(mbs["i_partition"], minlength=20)[[13,14,15,16]] if True else "")
