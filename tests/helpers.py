"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ANALYSIS_FIXTURES = ["qcif_hex_subme5", "qcif_dia_subme2", "qcif_umh_subme4_psub8", "qcif_esa_subme3",
                     "qcif_tesa_subme5_psub8", "qcif_hex_noisy_partitions", "cif_umh_subme5"]
# --subme 6 / 7 (RD mode decision): CABAC and CAVLC sizes, psy-RD on / off, embedding off (no P_8x8 then, analyse.c:2841)
RD_FIXTURES = ["qcif_hex_subme6", "qcif_umh_subme7_cavlc", "qcif_dia_subme6_nopsy_noisy", "qcif_esa_subme6_noembed", "cif_umh_subme7",
               # sub-8x8 partitions priced by x264_rd_cost_part (rdo.c:202-245)
               "qcif_hex_subme6_psub8", "qcif_hex_subme7_psub8_cavlc", "qcif_umh_subme6_psub8", "qcif_tesa_subme6"]


def fixture_params(g, make_params, **over):
    """the parameter block a fixture was minted with (older fixtures predate the RD fields)"""
    kw = dict(me=int(g["me"]), subme=int(g["subme"]), mv_range=int(g["mv_range"]), inter=int(g["inter"]) | 1,
              me_range=int(g["me_range"]), tscale=256)
    if "cabac" in g:
        kw.update(cabac=int(g["cabac"]), psy_rd=int(g["psy_rd_fix8"]) / 256.0, chroma_qp_offset=int(g["chroma_qp_offset"]))
    kw.update(over)
    return make_params(int(g["width"]), int(g["height"]), **kw)


# reference-harness field -> pcamv_mb_t field
FIELD_MAP = (("type", "i_type"), ("partition", "i_partition"), ("sub_partition", "i_sub_partition"),
             ("ref", "ref"), ("mv", "mv"), ("mv_stego", "mv_stego"), ("stego_cost", "inter_stego_cost"),
             ("pskip_mv", "pskip_mv"), ("mvr16", "mvr16"), ("used", "used"))
BX = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]
BY = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def compare_records(golden_mbs, got, what=""):
    """bit-exact comparison of every field of the pass-1 record"""
    for fr, fo in FIELD_MAP:
        a, b = golden_mbs[fr], got[fo]
        if not np.array_equal(a, b):
            bad = np.argwhere((a != b).reshape(len(a), -1).any(1)).ravel()
            raise AssertionError(f"{what}: field {fr} differs at MBs {bad[:8].tolist()}: "
                                 f"expected {a[bad[0]].tolist()} got {b[bad[0]].tolist()}")


def mv_field(mbs_mv, mbw, mbh):
    """the record's MVs (x264 block order per macroblock) as the frame's 4x4 motion field + its (all zero) reference field"""
    mv = np.asarray(mbs_mv, np.int16).reshape(mbh, mbw, 16, 2)
    mvf = np.zeros((mbh, 4, mbw, 4, 2), np.int16)
    for i in range(16):
        mvf[:, BY[i], :, BX[i]] = mv[:, :, i]
    return mvf.reshape(mbh * 4, mbw * 4, 2), np.zeros((mbh * 2, mbw * 2), np.int8)


def carrier_lsbs(mbs):
    """LSB(mvx+mvy) of every carrier MV in the order of encoder.c:1566-1647"""
    out = []
    for mb in mbs:
        if not mb["used"]:
            continue
        if mb["i_type"] == 5:
            for i in range(4):
                sp = mb["i_sub_partition"][i]
                slots = {3: [4 * i], 2: [4 * i, 4 * i + 1], 1: [4 * i, 4 * i + 2], 0: [4 * i + j for j in range(4)]}[int(sp)]
                out += [(int(mb["mv"][s][0]) + int(mb["mv"][s][1])) & 1 for s in slots]
        elif mb["i_type"] == 4:
            slots = {16: [0], 15: [0, 4], 14: [0, 8]}[int(mb["i_partition"])]
            out += [(int(mb["mv"][s][0]) + int(mb["mv"][s][1])) & 1 for s in slots]
    return np.array(out, np.uint8)
