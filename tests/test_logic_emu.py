"""The product's wave-uniform control code (csrc/pcamv_logic.h, pcamv_mbkernels.h), compiled for
the CPU with scalar primitives (tests/emu/), against the reference-minted fixtures.  This checks on
the CPU the same search / partition / RCA logic the HIP kernels run, including the anti-diagonal
processing order and the split into search, RCA and encode phases, and the dataflow schedule's
per-macroblock search -> RCA -> reconstruction sequence (order 2)."""
import numpy as np
import pytest

import helpers
import orc
from emu import emu

FIX = ["qcif_hex_subme5", "qcif_dia_subme2", "qcif_umh_subme4_psub8", "qcif_esa_subme3", "qcif_tesa_subme5_psub8", "qcif_hex_noisy_partitions", "cif_umh_subme5"]


@pytest.mark.parametrize("order", [1, 2], ids=["diagonal_phases", "dataflow_fused"])
@pytest.mark.parametrize("name", FIX)
def test_control_logic_matches_reference(name, order):
    g = helpers.load(name)
    W, H = int(g["width"]), int(g["height"])
    p = orc.make_params(W, H, me=int(g["me"]), subme=int(g["subme"]), mv_range=int(g["mv_range"]),
                        inter=int(g["inter"]), me_range=int(g["me_range"]), tscale=256)
    o = orc.Oracle(p)     # used only to produce the padded half-pel planes the kernels read
    for t in range(1, int(g["frames"]) + 1):
        prev = (g[f"f{t}_prev_mv"], g[f"f{t}_prev_ref"]) if f"f{t}_prev_mv" in g else (None, None)
        o.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev)
        mbs, rec = emu.analyse_pframe(orc, p, int(g["qp"]), 1, [g[f"f{t}_fenc_{c}"] for c in "yuv"], o.ref_planes(),
                                      g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev, diag=order)
        helpers.compare_records(g[f"f{t}_mbs"], mbs, f"{name} frame {t}")
        for k, nm in enumerate("yuv"):
            assert np.array_equal(rec[k], g[f"f{t}_rec_{nm}"]), f"{name} frame {t}: recon {nm}"
    o.close()


# --subme 6 / 7: the RD mode decision in the shared control code.  With CABAC the context states chain the macroblocks of a
# frame in raster order (order 3: raster, fused); CAVLC sizes depend on the left / top neighbours only, so the dataflow
# schedule's orders apply as well.
RD = [(n, o) for n in helpers.RD_FIXTURES for o in ((3,) if "cavlc" not in n else (1, 2, 3))]


@pytest.mark.parametrize("name,order", RD, ids=[f"{n}-order{o}" for n, o in RD])
def test_rd_mode_decision_logic_matches_reference(name, order):
    g = helpers.load(name)
    W, H = int(g["width"]), int(g["height"])
    p = helpers.fixture_params(g, orc.make_params)
    embed = int(g["embed"])
    o = orc.Oracle(p)     # used only to produce the padded half-pel planes the kernels read
    for t in range(1, int(g["frames"]) + 1):
        prev = (g[f"f{t}_prev_mv"], g[f"f{t}_prev_ref"]) if f"f{t}_prev_mv" in g else (None, None)
        o.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev)
        hashes = np.zeros(len(g[f"f{t}_mbs"]), np.uint32)
        mbs, rec = emu.analyse_pframe(orc, p, int(g["qp"]), embed, [g[f"f{t}_fenc_{c}"] for c in "yuv"], o.ref_planes(),
                                      g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev, diag=order, state_hash=hashes)
        if f"f{t}_cabac_state_hash" in g:
            bad = np.nonzero(hashes != g[f"f{t}_cabac_state_hash"])[0]
            assert len(bad) == 0, f"{name} frame {t}: CABAC context states differ from macroblock {bad[0]} on"
        helpers.compare_records(g[f"f{t}_mbs"], mbs, f"{name} frame {t}")
        for k, nm in enumerate("yuv"):
            assert np.array_equal(rec[k], g[f"f{t}_rec_{nm}"]), f"{name} frame {t}: recon {nm}"
    o.close()


def test_strip_layout_arithmetic():
    """Device layout of the luma reference planes (pcamv_common.h): the multiply-shift strip index is x / 28 for every x a
    padded plane can have (up to 8K widths and beyond), a 4-byte fetch at any x -- and at x + 1 -- stays inside one strip's row, distinct
    pixels of a plane (up to the four repeated columns) never share a byte, and every byte lies inside the plane's allocation."""
    import ctypes as C
    lib = C.CDLL(emu.build())
    lib.emu_strip_offset.restype = C.c_longlong
    lib.emu_strip_plane_size.restype = C.c_longlong
    assert lib.emu_strip_index_first_bad(40000) == -1
    for w, h in ((176, 144), (1920, 1088), (3840, 2160), (7680, 4320)):
        stride, lines = (w + 64 + 15) & ~15, h + 64
        psz = lib.emu_strip_plane_size(stride, lines)
        seen = {}
        for y in (0, 1, lines // 2, lines - 1):
            for x in range(stride):
                o = lib.emu_strip_offset(x, y, lines)
                assert 0 <= o and o + 4 <= psz
                assert o % 32 == x % 28 and o % 32 + 1 + 4 <= 32           # bytes x .. x + 4 (fetch at x + 1 included) in one 32-byte row
                assert seen.setdefault(o, (x, y)) == (x, y)
                if x % 28 < 4 and x >= 28:                                   # the repeat at the end of the strip before
                    assert lib.emu_strip_offset(x - 28, y, lines) + 28 == o - (32 * lines - 28)
