"""H.264 MV-syntax extractor (pcamv_gpu_parse_pslice_cabac / _cavlc: host code of the library, no GPU needed) -- the decode side of the BER
check.  Golden inputs are slices as the REFERENCE's own entropy coders (CABAC and CAVLC) wrote them (tests/golden/pslice_*.npz, minted by
oracle/gen_golden.py --pslice-only through the harness); the parser must read back exactly what the reference coded:
every macroblock's type, partition, sub-partitions and motion vectors (P_SKIP inferred, MV prediction of 8.4.1), for first-pass
frames with every partitioning incl. p4x4, and for FINAL frames (flipped MVs) the payload comes back out of the parsed motion
vectors with BER 0.  (Final frames only with 16x16 partitions: the gcc-built reference's second pass writes streams whose
16x8 / 8x16 / P_8x8 macroblocks a decoder reads differently from the encoder's own state -- stale i_partition in its MV
prediction)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))
import helpers  # noqa: E402

FIXTURES = ["pslice_qcif_hex_subme5_final", "pslice_cif_umh_subme7_final", "pslice_cif_umh_subme7_partitions",
            "pslice_qcif_hex_subme6_qp34", "pslice_cif_dia_subme4_p4x4_qp16",
            # --no-cabac: mb_skip_run, Exp-Golomb header, residual_block_cavlc (QP 10: level escapes)
            "pslice_cavlc_cif_umh_subme7_final", "pslice_cavlc_cif_hex_subme5_p4x4_qp10", "pslice_cavlc_qcif_hex_subme6_qp34"]


def _parse(g):
    import pcamv_amd
    if "cabac" in g and not int(g["cabac"]):
        return pcamv_amd.parse_pslice_cavlc(g["slice_data"].tobytes(), int(g["width"]) // 16, int(g["height"]) // 16)
    return pcamv_amd.parse_pslice_cabac(g["slice_data"].tobytes(), int(g["width"]) // 16, int(g["height"]) // 16, int(g["qp"]))


@pytest.mark.parametrize("name", FIXTURES)
def test_parser_reads_back_what_the_reference_coded(name):
    g = helpers.load(name)
    got = _parse(g)
    for a, b in (("type", "i_type"), ("partition", "i_partition"), ("sub_partition", "i_sub_partition"), ("mv", "mv")):
        bad = np.argwhere((g[a] != got[b]).reshape(len(got), -1).any(1)).ravel()
        assert len(bad) == 0, f"{a} differs at macroblocks {bad[:8].tolist()} ({len(bad)} in all)"
    assert (got["ref"] == 0).all()


@pytest.mark.parametrize("name", [n for n in FIXTURES if n.endswith("_final")])
def test_payload_comes_back_out_of_the_stream(name):
    import pcamv_amd
    g = helpers.load(name)
    got = _parse(g)
    got["used"] = g["used"]             # which macroblocks carry: every coded (not skipped) one (encoder.c:1566)
    assert np.array_equal(got["used"] != 0, got["i_type"] != pcamv_amd.P_SKIP)
    lsb = helpers.carrier_lsbs(got)
    assert len(lsb) == int(g["n"])
    msg = pcamv_amd.stc_extract(lsb, int(g["m"]))
    assert np.array_equal(msg, g["message"]), "decode-side BER != 0"


@pytest.mark.parametrize("name", FIXTURES)
def test_slice_data_inside_a_nal_unit_behind_a_header(name):
    """the NAL layer and an unaligned slice header in front: the NAL unit was escaped by the reference's own x264_nal_encode
    (common/common.c:658; 00 00 03 inserted in the zero-filled stand-in header of the CABAC fixtures, and 6 times inside the
    QP-10 CAVLC slice), the slice data starts at bit nal_hdr_bits of the RBSP (CABAC: after the alignment ones)"""
    import pcamv_amd
    g = helpers.load(name)
    nal = g["nal"].tobytes()
    assert nal[:4] == b"\x00\x00\x00\x01"
    rbsp, ref_idc, typ = pcamv_amd.nal_to_rbsp(nal)
    assert (ref_idc, typ) == (2, 1) and len(nal) - 5 - len(rbsp) == int(g["nal_escapes"])
    assert pcamv_amd.nal_to_rbsp(nal[4:])[0] == rbsp and pcamv_amd.nal_to_rbsp(nal[1:])[0] == rbsp       # no / short start code
    cabac = not ("cabac" in g and not int(g["cabac"]))
    got = pcamv_amd.parse_pslice_at(rbsp, int(g["nal_hdr_bits"]), int(g["width"]) // 16, int(g["height"]) // 16, int(g["qp"]) if cabac else None)
    for a, b in (("type", "i_type"), ("partition", "i_partition"), ("sub_partition", "i_sub_partition"), ("mv", "mv")):
        assert np.array_equal(g[a], got[b]), a
    with pytest.raises(pcamv_amd.PcamvError):           # one bit off: an error, never a crash (CABAC: the alignment bits are checked)
        off = pcamv_amd.parse_pslice_at(rbsp, int(g["nal_hdr_bits"]) - (3 if cabac else 1), int(g["width"]) // 16, int(g["height"]) // 16, int(g["qp"]) if cabac else None)
        if np.array_equal(off["mv"], g["mv"]):
            raise AssertionError("a shifted start parsed to the same motion")
        raise pcamv_amd.PcamvError("different motion")
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.nal_to_rbsp(b"\x00\x00\x01\x41\x12\x00\x00\x01\x33")      # a start code inside the unit


def test_skip_run_beyond_the_picture_is_reported():
    """CAVLC: an mb_skip_run that claims more macroblocks than the picture has left (ue(12) = 0001101, then the trailing bit)"""
    import pcamv_amd
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.parse_pslice_cavlc(bytes([0b00011011, 0b00000000]), 3, 3)
    got = pcamv_amd.parse_pslice_cavlc(bytes([0b00010101]), 3, 3)             # ue(9) = 0001010 + trailing 1: nine skipped macroblocks
    assert (got["i_type"] == pcamv_amd.P_SKIP).all()


def test_damaged_streams_are_reported():
    import pcamv_amd
    g = helpers.load("pslice_qcif_hex_subme5_final")
    data = g["slice_data"].tobytes()
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.parse_pslice_cabac(data[:len(data) // 2], 11, 9, 26)         # truncated: runs out before the last macroblock
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.parse_pslice_cabac(data, 11, 8, 26)                          # wrong picture size: end_of_slice in the wrong place
    v = helpers.load("pslice_cavlc_qcif_hex_subme6_qp34")["slice_data"].tobytes()
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.parse_pslice_cavlc(v[:len(v) // 2], 11, 9)
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.parse_pslice_cavlc(v, 11, 8)                                 # data left after the last macroblock
    bad = bytearray(data); bad[40] ^= 0x55
    try:                                                                        # a flipped byte: an error, or different motion -- never a crash
        got = pcamv_amd.parse_pslice_cabac(bytes(bad), 11, 9, 26)
        assert not np.array_equal(got["mv"], g["mv"])
    except pcamv_amd.PcamvError:
        pass


def test_live_against_the_reference_coder():
    """wherever the reference harness is built: more frames, chained (the second P frame predicts from the first one's motion)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import refh
    if not refh.available():
        pytest.skip("oracle/_ref/libpcamv_ref.so not built (needs /root/reference)")
    import orc
    import pcamv_amd
    from pcamv_amd.synth import make_clip
    for (W, H, me, subme, qp, inter, seed, static, noise) in [(320, 240, "hex", 6, 20, 0x11, 31, 64, 20), (176, 144, "umh", 5, 40, 0x31, 32, 0, 35),
                                                              (352, 288, "esa", 3, 12, 0x31, 33, 96, 30)]:
        clip = make_clip(W, H, 3, seed=seed, static_cols=static, noise=noise)
        for cabac in (1, 0):
            r = refh.Ref(W, H, qp=qp, me=me, subme=subme, mv_range=orc.level_mv_range(W, H), cabac=cabac, embed=1, inter_flags=inter)
            ref, prev = clip[0], (None, None)
            for t in (1, 2):
                r.set_ref(*ref, *prev); r.set_fenc(*clip[t])
                mbs, rec = r.analyse_pframe(qp)
                got = pcamv_amd.parse_pslice_cabac(r.slice_data(), W // 16, H // 16, qp) if cabac else pcamv_amd.parse_pslice_cavlc(r.slice_data(), W // 16, H // 16)
                for a, b in (("type", "i_type"), ("partition", "i_partition"), ("sub_partition", "i_sub_partition"), ("mv", "mv")):
                    assert np.array_equal(mbs[a], got[b]), (W, H, me, cabac, t, a)
                ref, prev = rec, helpers.mv_field(mbs["mv"], W // 16, H // 16)


def test_parsers_survive_arbitrary_input():
    """the extractor takes bytes from outside: random strings and randomly damaged real slices must come back as an error or as
    some motion field, never as a crash or a hang (every loop of the parsers is bounded, every read is checked against the end)"""
    import pcamv_amd
    rng = np.random.default_rng(2024)
    real = {True: helpers.load("pslice_qcif_hex_subme6_qp34")["slice_data"].tobytes(), False: helpers.load("pslice_cavlc_qcif_hex_subme6_qp34")["slice_data"].tobytes()}
    outcomes = {"ok": 0, "error": 0}
    for k in range(300):
        cabac = bool(k & 1)
        if k % 3 == 0:
            data = rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8).tobytes()
        else:
            d = bytearray(real[cabac])
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(0, len(d)))] ^= int(rng.integers(1, 256))
            if k % 5 == 0:
                d = d[:int(rng.integers(1, len(d)))]
            data = bytes(d)
        try:
            got = pcamv_amd.parse_pslice_cabac(data, 11, 9, int(rng.integers(0, 52))) if cabac else pcamv_amd.parse_pslice_cavlc(data, 11, 9)
            assert len(got) == 99
            outcomes["ok"] += 1
        except pcamv_amd.PcamvError:
            outcomes["error"] += 1
    assert outcomes["error"] > 100, outcomes


def test_parsers_under_sanitizers(tmp_path):
    """the same parsers compiled for the CPU with -fsanitize=address,undefined (tests/fuzz/fuzz_mvsyntax.cpp): damaged real slices
    in exact-size heap buffers; any out-of-bounds read, overflow or undefined shift aborts the run"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_mvsyntax")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "video-steganography-pcamv_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "fuzz", "fuzz_mvsyntax.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    args = [exe, "1500"]
    for name in ("pslice_qcif_hex_subme6_qp34", "pslice_cavlc_qcif_hex_subme6_qp34", "pslice_cif_dia_subme4_p4x4_qp16", "pslice_cavlc_cif_hex_subme5_p4x4_qp10"):
        g = helpers.load(name)
        f = tmp_path / (name + ".bin")
        f.write_bytes(g["slice_data"].tobytes())
        args += [str(f), str(int(g["width"]) // 16), str(int(g["height"]) // 16), str(int(g["qp"])), str(int(g["cabac"]) if "cabac" in g else 1)]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "err" in r.stdout
