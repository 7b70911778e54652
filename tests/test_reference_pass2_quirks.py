"""What the reference's second pass (analyse.c:2870-3107, firstTime == 0) does when its own sources are
compiled here, recorded as tests because the next row of the build (pass-2 reconstruction + loop filter on
the GPU) has to decide what "parity" means for it:

* with only 16x16 partitions and no flips, pass 2 reproduces pass 1 exactly (forced type + MVs from
  h->info.cache), so "final MVs = pass-1 record + flips" (pcamv_gpu_final_mvs) is the reference's result;
* the pass-1 record loop `cache.mv[idx] = ...[x264_scan8[idx++]]` (analyse.c:3535-3540, 3625-3630) modifies
  and uses idx unsequenced; built with gcc the stores land one slot late, so pass 2 reads the FIRST
  partition's MV for the second partition of 16x8 / 8x16 macroblocks (mv[8] / mv[4]).  The product keeps the
  evident intent (slot k = block k).

Skipped where oracle/_ref/libpcamv_ref.so is absent."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import orc  # noqa: E402
import refh  # noqa: E402

pytestmark = pytest.mark.skipif(not refh.available(), reason="oracle/_ref/libpcamv_ref.so not built (needs /root/reference)")


def _run(inter):
    from pcamv_amd.synth import make_clip
    W, H = 176, 144
    clip = make_clip(W, H, 2, seed=5, static_cols=48)
    r = refh.Ref(W, H, qp=26, me="hex", subme=5, mv_range=orc.level_mv_range(W, H), embed=1, inter_flags=inter | 0x1 | 0x100)
    r.set_ref(*clip[0]); r.set_fenc(*clip[1])
    mbs, rec = r.analyse_pframe()
    n = 0
    for mb in mbs:
        if mb["used"]:
            n += 1 if mb["partition"] == 16 else 2
    m2, nnz, rec2, dbk, walked = r.pass2_pframe(np.zeros(n, np.int8))
    return mbs, rec, m2, rec2, dbk, walked, n


def test_pass2_without_flips_reproduces_pass1_for_16x16_only():
    mbs, rec, m2, rec2, dbk, walked, n = _run(0)
    assert walked == n
    assert set(np.unique(mbs["partition"])) <= {16}
    assert np.array_equal(m2["type"], mbs["type"]) and np.array_equal(m2["mv"], mbs["mv"])
    for a, b in zip(rec, rec2):
        assert np.array_equal(a, b)
    assert any((a != b).any() for a, b in zip(rec2, dbk)), "the loop filter changed nothing"


def test_gcc_build_of_the_record_loop_shifts_second_partitions():
    mbs, rec, m2, rec2, dbk, walked, n = _run(0x10)
    two = np.argwhere(np.isin(mbs["partition"], (14, 15)) & (mbs["type"] == 4)).ravel()
    assert len(two) > 0 and walked == n
    diff = [xy for xy in two if not np.array_equal(mbs["mv"][xy], m2["mv"][xy])]
    assert diff, "expected the unsequenced idx++ of the reference's record loop to show (gcc)"
    for xy in diff:
        assert (m2["mv"][xy] == m2["mv"][xy][0]).all()        # second partition took the first one's MV
