"""The library's syndrome-trellis extractor is host code (no GPU needed): message bits embedded by the oracle's stc_embed
(pinned on the reference's embed.h by tests/golden/primitives.npz and the live harness tests) must come back out of the stego
bits, for the tabulated sub-matrix widths AND for the ones whose columns the reference draws from its process-wide LCG
(widths 1 and 21..256, embed.h:134-199) -- there the extractor is handed the generator's state and carries it from frame to
frame like the reference's extractor process does."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))


def _frames(rng, shapes):
    for n, m in shapes:
        cover = rng.integers(0, 2, n).astype(np.uint8)
        rho = (rng.random(n) * 40 + 1).astype(np.float32)
        msg = rng.integers(0, 2, m).astype(np.uint8)
        yield cover, rho, msg


def test_extractor_carries_the_column_generator_across_frames():
    import orc
    import pcamv_amd
    orc.lib().orc_stc_lcg_reset(1)
    lcg = pcamv_amd.StcLcg(1)
    rng = np.random.default_rng(35)
    # (carriers, message bits): widths 26|27 (LCG), 11|12 (tables), 20|21 (table + LCG), 64 (LCG), 1 (m = n: LCG), 256 (the widest allowed)
    shapes = [(935, 35), (400, 35), (717, 35), (640, 10), (48, 48), (2560, 10), (300, 12)]
    for cover, rho, msg in _frames(rng, shapes):
        ok, stego = orc.stc_embed(cover, msg, rho)
        assert ok == 1
        got = pcamv_amd.stc_extract(stego, len(msg), lcg=lcg)
        assert np.array_equal(got, msg), (len(cover), len(msg))
    assert lcg.state.value != 1, "none of the frames used the generator"


def test_plain_extractor_refuses_untabulated_widths():
    import pcamv_amd
    stego = np.zeros(935, np.uint8)
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.stc_extract(stego, 35)                      # width 26 | 27 without the generator's state
    assert pcamv_amd.stc_extract(stego, 85).sum() == 0        # width 11: tables
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.stc_extract(stego, 3, lcg=pcamv_amd.StcLcg(1))   # width 311 > 256: the embedder fails too (embed.h:286)
