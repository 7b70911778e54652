"""The CPU restatement against the reference's own code, live (oracle/_ref/libpcamv_ref.so built from
/root/reference by oracle/Makefile): option and quantiser ranges beyond the committed fixtures, so that the
oracle the GPU sweep test trusts is pinned there too.  Skipped where the reference library is absent
(the GPU box has no /root/reference; the library travels only if it was built here)."""
import os
import sys

import numpy as np
import pytest

import helpers
import orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import refh  # noqa: E402

pytestmark = pytest.mark.skipif(not refh.available(), reason="oracle/_ref/libpcamv_ref.so not built (needs /root/reference)")

LIVE = [
    # (W, H, me, me_range, subme, inter, qp, seed, static_cols)
    (176, 144, "dia", 16, 1, 0x10, 12, 41, 48),
    (176, 144, "hex", 16, 3, 0x10, 20, 42, 32),
    (176, 144, "umh", 24, 5, 0x10, 35, 43, 0),
    (176, 144, "umh", 8, 4, 0x30, 45, 44, 64),
    (176, 144, "esa", 8, 2, 0x10, 28, 46, 48),
    (176, 144, "esa", 16, 5, 0x30, 33, 48, 32),
    (176, 144, "tesa", 16, 4, 0x10, 30, 49, 32),
]


@pytest.mark.parametrize("cfg", LIVE, ids=[f"{c[2]}_r{c[3]}_s{c[4]}_i{c[5]:x}_qp{c[6]}" for c in LIVE])
def test_oracle_matches_reference_code(cfg):
    from pcamv_amd.synth import make_clip
    W, H, me, me_range, subme, inter, qp, seed, static = cfg
    clip = make_clip(W, H, 3, seed=seed, static_cols=static)
    mvr = orc.level_mv_range(W, H)
    r = refh.Ref(W, H, qp=qp, me=me, subme=subme, mv_range=mvr, embed=1, inter_flags=inter | 0x1 | 0x100, me_range=me_range)
    o = orc.Oracle(orc.make_params(W, H, me=me, me_range=me_range, subme=subme, mv_range=mvr, inter=inter))
    ref, prev = clip[0], (None, None)
    for t in (1, 2):
        if prev[0] is None:
            r.set_ref(*ref)
        else:
            r.set_ref(*ref, prev_mv=prev[0], prev_ref=prev[1])
        r.set_fenc(*clip[t])
        o.set_ref(*ref, *prev); o.set_fenc(*clip[t])
        mbs_r, rec_r = r.analyse_pframe()
        mbs_o, rec_o = o.analyse_pframe(qp, 1)
        helpers.compare_records(mbs_r, mbs_o, f"{cfg} frame {t}")
        for a, b in zip(rec_r, rec_o):
            assert np.array_equal(a, b), f"{cfg} frame {t}: reconstruction"
        prev = helpers.mv_field(mbs_o["mv"], W // 16, H // 16)
        ref = rec_o
    o.close()


# --subme 6 / 7: RD mode decision with the size-only CABAC / CAVLC coders, psy-RD, intra SATD thresholds
LIVE_RD = [
    # (W, H, me, subme, qp, seed, static_cols, cabac, psy_rd, noise, embed)
    (176, 144, "hex", 6, 12, 51, 32, 1, 1.0, 40, 1),
    (176, 144, "hex", 7, 20, 52, 32, 0, 1.0, 40, 1),
    (176, 144, "umh", 6, 38, 53, 0, 1, 2.0, 25, 1),
    (176, 144, "dia", 7, 45, 54, 48, 0, 0.1, 6, 1),
    (176, 144, "esa", 6, 30, 55, 16, 1, 0.0, 40, 0),
    (352, 288, "hex", 6, 28, 56, 64, 0, 1.0, 12, 1),
]


@pytest.mark.parametrize("cfg", LIVE_RD, ids=[f"{c[2]}_s{c[3]}_qp{c[4]}_{'cabac' if c[7] else 'cavlc'}_psy{c[8]}_e{c[10]}" for c in LIVE_RD])
def test_oracle_rd_mode_decision_matches_reference_code(cfg):
    from pcamv_amd.synth import make_clip
    W, H, me, subme, qp, seed, static, cabac, psy, noise, embed = cfg
    clip = make_clip(W, H, 3, seed=seed, static_cols=static, noise=noise)
    mvr = orc.level_mv_range(W, H)
    r = refh.Ref(W, H, qp=qp, me=me, subme=subme, mv_range=mvr, embed=embed, inter_flags=0x111, cabac=cabac, psy_rd=psy)
    o = orc.Oracle(orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=0x11, cabac=cabac, psy_rd=psy))
    hr, ho = r.debug_state_hash(), o.debug_state_hash()
    ref, prev = clip[0], (None, None)
    for t in (1, 2):
        if prev[0] is None:
            r.set_ref(*ref)
        else:
            r.set_ref(*ref, prev_mv=prev[0], prev_ref=prev[1])
        r.set_fenc(*clip[t])
        o.set_ref(*ref, *prev); o.set_fenc(*clip[t])
        mbs_r, rec_r = r.analyse_pframe()
        mbs_o, rec_o = o.analyse_pframe(qp, embed)
        helpers.compare_records(mbs_r, mbs_o, f"{cfg} frame {t}")
        for a, b in zip(rec_r, rec_o):
            assert np.array_equal(a, b), f"{cfg} frame {t}: reconstruction"
        if cabac:
            assert np.array_equal(hr, ho), f"{cfg} frame {t}: CABAC context states"
        prev = helpers.mv_field(mbs_o["mv"], W // 16, H // 16)
        ref = rec_o
    o.close()
