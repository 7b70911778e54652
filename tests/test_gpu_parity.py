"""Parity tests proper: the HIP path through the C ABI (libpcamv_gpu.so) against (a) the fixtures
minted from the reference's own C code and (b) the CPU oracle on the same seeded inputs.
Bit-exact everywhere: MVs, partitions, SAD/SATD scores, RCA costs, reconstruction, planes,
cover / price / stego / flip vectors.  Run with -m gpu on the MI355X box."""
import hashlib

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def pc():
    import torch
    torch.cuda.init()                   # the tests that hand device tensors to the library: torch's lazy HIP initialisation, done late
    #                                     (after the library's own HIP calls, in some test orders), found "No HIP GPUs"
    import pcamv_amd
    pcamv_amd.load_library()            # fails loudly if the HIP library is missing
    return pcamv_amd


def _params(pc, W, H, me, subme, inter, mv_range, me_range=16, tscale=256, cabac=1, psy_fix8=None, chroma_qp_offset=None):
    p = pc.param_default(W, H)
    pc.param_parse(p, "subme", subme)       # also settles psy-RD / the chroma QP offset the way x264_validate_parameters does
    p.i_me_method, p.inter, p.i_mv_range, p.i_me_range, p.i_tscale, p.b_cabac = me, inter | 1, mv_range, me_range, tscale, cabac
    if psy_fix8 is not None:
        p.i_psy_rd, p.i_chroma_qp_offset = psy_fix8, chroma_qp_offset
    return p


def _fixture_params(pc, g):
    kw = {}
    if "cabac" in g:
        kw = dict(cabac=int(g["cabac"]), psy_fix8=int(g["psy_rd_fix8"]), chroma_qp_offset=int(g["chroma_qp_offset"]))
    return _params(pc, int(g["width"]), int(g["height"]), int(g["me"]), int(g["subme"]), int(g["inter"]), int(g["mv_range"]), int(g["me_range"]), **kw)


GPU_FIXTURES = ["qcif_hex_subme5", "qcif_dia_subme2", "qcif_umh_subme4_psub8", "qcif_esa_subme3", "qcif_tesa_subme5_psub8", "qcif_hex_noisy_partitions", "cif_umh_subme5"]


@pytest.mark.parametrize("name", GPU_FIXTURES + helpers.RD_FIXTURES)
def test_pframe_analysis_matches_reference_fixture(pc, name):
    """every field of the pass-1 record, the reconstruction and the half-pel planes against what the reference's own code
    computed; for --subme 6 / 7 with CABAC also the context states after every macroblock"""
    g = helpers.load(name)
    W, H = int(g["width"]), int(g["height"])
    embed = int(g["embed"]) if "embed" in g else 1
    enc = pc.Encoder(_fixture_params(pc, g))
    want_hash = "f1_cabac_state_hash" in g
    if want_hash:
        enc.debug_state_hash(True)
    for t in range(1, int(g["frames"]) + 1):
        prev = (g[f"f{t}_prev_mv"], g[f"f{t}_prev_ref"]) if f"f{t}_prev_mv" in g else (None, None)
        enc.set_ref(g[f"f{t}_ref_y"], g[f"f{t}_ref_u"], g[f"f{t}_ref_v"], *prev)
        enc.upload_fenc(g[f"f{t}_fenc_y"], g[f"f{t}_fenc_u"], g[f"f{t}_fenc_v"])
        planes = enc.ref_planes()
        for k in range(4):
            assert sha(planes[k]) == str(g[f"f{t}_plane_sha"][k]), f"{name} frame {t}: half-pel plane {k}"
        mbs, rec = enc.analyse_pframe(int(g["qp"]), embed=embed)
        if want_hash:
            bad = np.nonzero(enc.state_hash_fetch() != g[f"f{t}_cabac_state_hash"])[0]
            assert len(bad) == 0, f"{name} frame {t}: CABAC context states differ from macroblock {bad[0]} on ({len(bad)} in all)"
        helpers.compare_records(g[f"f{t}_mbs"], mbs, f"{name} frame {t}")
        for k, nm in enumerate("yuv"):
            assert np.array_equal(rec[k], g[f"f{t}_rec_{nm}"]), f"{name} frame {t}: recon {nm}"
    enc.close()


def test_block_costs_match_oracle(pc):
    """SAD / SATD / quarter-pel fetch / chroma MC of all 7 block sizes at random MVs."""
    import ctypes as C
    import orc
    from pcamv_amd.synth import make_clip
    W, H = 176, 144
    clip = make_clip(W, H, 2, seed=21)
    p = _params(pc, W, H, 1, 5, 0x10, 64)
    enc = pc.Encoder(p)
    enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1])
    o = orc.Oracle(orc.make_params(W, H, mv_range=64))
    o.set_ref(*clip[0])
    planes = o.ref_planes(); st = planes.shape[2]
    cu = np.ascontiguousarray(np.pad(clip[0][1], 16, mode="edge")); cv = np.ascontiguousarray(np.pad(clip[0][2], 16, mode="edge"))
    rng = np.random.default_rng(3)
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    offs = {0: [(0, 0)], 1: [(0, 0), (0, 8)], 2: [(0, 0), (8, 0)], 3: [(0, 0), (8, 0), (0, 8), (8, 8)],
            4: [(0, 4), (8, 12)], 5: [(4, 0), (12, 8)], 6: [(12, 12), (4, 8)]}
    req = []
    for i in range(600):
        ip = i % 7
        xo, yo = offs[ip][i % len(offs[ip])]
        req.append([int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16)), ip, xo, yo,
                    int(rng.integers(-40, 40)), int(rng.integers(-40, 40)), i % 2])
    got = enc.block_costs(26, req)
    L = orc.lib()
    fy, fu, fv = [np.ascontiguousarray(a) for a in clip[1]]
    for (mbx, mby, ip, xo, yo, mx, my, satd), g3 in zip(req, got):
        w, h = sizes[ip]
        fn = L.orc_satd if satd else L.orc_sad
        src = (C.c_void_p * 4)(*[planes[k].ctypes.data + (32 + mby * 16 + yo) * st + 32 + mbx * 16 + xo for k in range(4)])
        dst = np.zeros((h, w), np.uint8)
        L.orc_mc_luma(dst.ctypes.data_as(C.c_void_p), w, src, st, mx, my, w, h)
        e = C.c_void_p(fy.ctypes.data + (mby * 16 + yo) * W + mbx * 16 + xo)
        assert fn(ip, e, W, dst.ctypes.data_as(C.c_void_p), w) == g3[0], ("luma", mbx, mby, ip, xo, yo, mx, my, satd)
        if ip <= 3:
            for pl, (cpl, fpl) in enumerate(((cu, fu), (cv, fv))):
                d2 = np.zeros((h // 2, w // 2), np.uint8)
                L.orc_mc_chroma(d2.ctypes.data_as(C.c_void_p), w // 2,
                                C.c_void_p(cpl.ctypes.data + (16 + mby * 8 + yo // 2) * cpl.shape[1] + 16 + mbx * 8 + xo // 2),
                                cpl.shape[1], mx, my, w // 2, h // 2)
                e2 = C.c_void_p(fpl.ctypes.data + (mby * 8 + yo // 2) * (W // 2) + mbx * 8 + xo // 2)
                assert fn(ip + 3, e2, W // 2, d2.ctypes.data_as(C.c_void_p), w // 2) == g3[1 + pl], ("chroma", pl, ip, mx, my, satd)
    # batched evaluation (4 candidates per wavefront): answers must equal the single-candidate ones
    breq = [[r[0], r[1], r[2], r[3], r[4], r[5], r[6], (r[7] & 1) | 2] for r in req[:300]]
    bgot = enc.block_costs(26, breq)
    sreq = []
    for r in breq:
        for dx, dy in ((1, -1), (-2, 3), (3, 2)):
            sreq.append([r[0], r[1], r[2], r[3], r[4], r[5] + dx, r[6] + dy, r[7] & 1])
    sgot = enc.block_costs(26, sreq)[:, 0].reshape(-1, 3)
    assert np.array_equal(bgot, sgot), np.argwhere(bgot != sgot)[:5]
    enc.close(); o.close()


def _probe_req(a, b, top=None, left=None, avail=3):
    """one request of pcamv_gpu_rd_probe: a / b = (Y 16x16, U 8x8, V 8x8); top [3][28], left [3][16]"""
    r = np.zeros(1024, np.uint8)
    for k, blk in enumerate((a, b)):
        m = np.zeros((24, 16), np.uint8)
        m[:16] = blk[0]; m[16:, :8] = blk[1]; m[16:, 8:] = blk[2]
        r[384 * k:384 * (k + 1)] = m.ravel()
    if top is not None:
        r[768:852] = np.asarray(top, np.uint8).ravel()
        r[852:900] = np.asarray(left, np.uint8).ravel()
    r[900:904] = np.frombuffer(np.int32(avail).tobytes(), np.uint8)
    return r


def test_rd_metrics_and_intra_predictors_match_reference_vectors(pc):
    """SURVEY a3 and the intra predictors, probed directly (pcamv_gpu_rd_probe): ssd 16x16 / 8x8, hadamard_ac 16x16 against the
    vectors the reference's own pixel.c produced (tests/golden/primitives_rd.npz hac_res), the psy-RD energies and ssd_mb's psy
    term against the oracle's sa8d / satd / hadamard_ac (pinned on the same file), every 16x16 / chroma / 4x4 predictor of
    common/predict.c through its SATD (and, at subme 1, SAD) against the pixels the reference's predictors wrote (ipred_out)."""
    import ctypes as C
    import orc
    g = helpers.load("primitives_rd")
    L = orc.lib()
    L.orc_hadamard_ac.restype = C.c_uint64
    rng = np.random.default_rng(9)
    qp = 26
    enc = pc.Encoder(_params(pc, 176, 144, 1, 6, 0x10, 64))          # subme 6: SATD, psy-RD 1.0
    enc1 = pc.Encoder(_params(pc, 176, 144, 1, 1, 0x10, 64))         # subme 1: the intra costs are SADs
    lam = [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91][qp]

    def ptr(a):
        return C.c_void_p(a.ctypes.data)

    # ---- ssd / hadamard_ac / psy energies
    reqs, exp = [], []
    for i in range(len(g["hac_pix"])):
        by = np.ascontiguousarray(g["hac_pix"][i][:, :16])                       # the block whose hadamard_ac the reference computed
        ay = np.ascontiguousarray(g["hac_other"][i])
        au, av, bu, bv = [rng.integers(0, 256, (8, 8), dtype=np.uint8) for _ in range(4)]
        if i % 3 == 0:
            bu, bv = au.copy(), av.copy()
        reqs.append(_probe_req((ay, au, av), (by, bu, bv)))
        ssd = L.orc_ssd(0, ptr(ay), 16, ptr(by), 16) + L.orc_ssd(3, ptr(au), 8, ptr(bu), 8) + L.orc_ssd(3, ptr(av), 8, ptr(bv), 8)
        assert ssd == int(((ay.astype(int) - by) ** 2).sum() + ((au.astype(int) - bu) ** 2).sum() + ((av.astype(int) - bv) ** 2).sum())
        zero = np.zeros((16, 16), np.uint8)
        satd_sum = sum(L.orc_satd(6, ptr(zero), 16, C.c_void_p(ay.ctypes.data + 4 * y * 16 + 4 * x), 16) -
                       (L.orc_sad(6, ptr(zero), 16, C.c_void_p(ay.ctypes.data + 4 * y * 16 + 4 * x), 16) >> 1) for y in range(4) for x in range(4))
        sa8d_sum = sum(L.orc_sa8d(3, ptr(zero), 16, C.c_void_p(ay.ctypes.data + 8 * y * 16 + 8 * x), 16) -
                       (L.orc_sad(3, ptr(zero), 16, C.c_void_p(ay.ctypes.data + 8 * y * 16 + 8 * x), 16) >> 2) for y in range(2) for x in range(2))
        h4, h8 = int(g["hac_res"][i, 0, 0]), int(g["hac_res"][i, 0, 1])        # reference-minted
        acs = L.orc_hadamard_ac(0, ptr(by), 16)
        assert (acs & 0xffffffff, acs >> 32) == (h4, h8)
        psy = (abs(h4 - satd_sum) + abs(h8 - sa8d_sum)) >> 1
        exp.append((ssd, ssd + ((psy * 256 * lam + 128) >> 8), h4, h8, satd_sum, sa8d_sum))
    got = enc.rd_probe(qp, np.stack(reqs))
    for i, e in enumerate(exp):
        assert tuple(int(v) for v in got[i, :6]) == e, (i, got[i, :6].tolist(), e)

    # ---- intra predictors: mode m of kind k on the borders of ipred_in, scored against a random source block
    need = {0: {0: 3, 1: 3, 2: 3, 3: 3, 4: 1, 5: 2, 6: 0}, 1: {0: 3, 1: 3, 2: 3, 3: 3, 4: 1, 5: 2, 6: 0}}
    slot = {0: {0: 6, 1: 7, 2: 8, 3: 9, 4: 8, 5: 8, 6: 8}, 1: {0: 10, 1: 11, 2: 12, 3: 13, 4: 10, 5: 10, 6: 10}}
    reqs, chk = [], []
    for buf, outp, (kind, mode) in zip(g["ipred_in"], g["ipred_out"], g["ipred_kind"]):
        kind, mode = int(kind), int(mode)
        fy = rng.integers(0, 256, (16, 16), dtype=np.uint8); fu = rng.integers(0, 256, (8, 8), dtype=np.uint8); fv = rng.integers(0, 256, (8, 8), dtype=np.uint8)
        top = np.zeros((3, 28), np.uint8); left = np.zeros((3, 16), np.uint8)
        n = (16, 8, 4)[kind]
        planes = (0,) if kind != 1 else (1, 2)
        for c in planes:
            top[c, 3] = buf[7, 7]; top[c, 4:4 + (24 if kind == 0 else 8)] = buf[7, 8:8 + (24 if kind == 0 else 8)]
            left[c, :n] = buf[8:8 + n, 7]
        pred = np.ascontiguousarray(outp[8:8 + n, 8:8 + n])
        avail = need[kind][mode] if kind < 2 else 3
        reqs.append(_probe_req((fy, fu, fv), (fy, fu, fv), top, left, avail))
        chk.append((kind, mode, fy, fu, fv, pred))
    for e, fn in ((enc, L.orc_satd), (enc1, L.orc_sad)):
        got = e.rd_probe(qp, np.stack(reqs))
        for i, (kind, mode, fy, fu, fv, pred) in enumerate(chk):
            if kind == 0:
                want, have = fn(0, ptr(fy), 16, ptr(pred), 16), int(got[i, slot[0][mode]])
            elif kind == 1:
                want, have = fn(3, ptr(fu), 8, ptr(pred), 8) + fn(3, ptr(fv), 8, ptr(pred), 8), int(got[i, slot[1][mode]])
            else:
                blk = np.ascontiguousarray(fy[:4, :4])
                want, have = fn(6, ptr(blk), 4, ptr(pred), 4), int(got[i, 14 + mode])
            assert want == have, (("satd", "sad")[fn is L.orc_sad], kind, mode, i, want, have)
    enc.close(); enc1.close()


RD_SWEEP = [
    # (W, H, me, subme, qp, seed, static_cols, cabac, psy_rd, noise, embed)
    (176, 144, "hex", 6, 12, 51, 32, 1, 1.0, 40, 1),       # low QP: long level prefixes / escapes
    (176, 144, "hex", 7, 20, 52, 32, 0, 1.0, 40, 1),       # CAVLC, noisy: many coded blocks, partitions
    (176, 144, "umh", 6, 38, 53, 0, 1, 2.0, 25, 1),        # strong psy-RD, coarse quantiser
    (176, 144, "dia", 7, 45, 54, 48, 0, 0.1, 6, 1),        # psy-RD below 0.25 (chroma QP offset -1), CAVLC
    (176, 144, "esa", 6, 30, 55, 16, 1, 0.0, 40, 0),       # embedding off: no P_8x8 from the RD stage (analyse.c:2841)
    (352, 288, "hex", 6, 28, 56, 64, 0, 1.0, 12, 1),
    (320, 240, "umh", 7, 26, 57, 96, 1, 1.0, 6, 1),        # config 3's options (--me umh --subme 7) on a wider picture
    (1280, 720, "hex", 6, 26, 11, 320, 1, 1.0, 6, 1),      # 720p: the raster chain over 3600 macroblocks
    # sub-8x8 partitions priced by x264_rd_cost_part (rdo.c:202-245; a 12th field: the inter flags)
    (176, 144, "hex", 6, 26, 61, 0, 1, 1.0, 30, 1, 0x30),
    (176, 144, "umh", 7, 18, 68, 32, 0, 1.0, 40, 1, 0x30),   # CAVLC: non-zero COUNTS left over between the trials
    (320, 240, "dia", 6, 32, 69, 64, 1, 0.5, 25, 1, 0x30),
    (64, 48, "hex", 6, 24, 70, 0, 1, 1.0, 35, 1, 0x30),      # 4 macroblocks wide: too narrow for the speculative chain (plain one-wave-per-SIMD build)
    # --me tesa at the RD levels (pcamv_rd_tesa.hip): CAVLC (wavefront order), and with sub-8x8 partitions
    (176, 144, "tesa", 7, 30, 82, 32, 0, 1.0, 30, 1),
    (176, 144, "tesa", 6, 22, 83, 0, 1, 1.0, 30, 1, 0x30),
]


@pytest.mark.parametrize("cfg", RD_SWEEP, ids=[f"{c[0]}x{c[1]}_{c[2]}_s{c[3]}_qp{c[4]}_{'cabac' if c[7] else 'cavlc'}_psy{c[8]}_e{c[10]}{'_p4x4' if len(c) > 11 else ''}" for c in RD_SWEEP])
def test_rd_mode_decision_matches_oracle(pc, cfg):
    """--subme 6 / 7 (the reference's default): intra SATD thresholds, psy-RD, size-only CABAC / CAVLC, context adaptation --
    record, reconstruction and (CABAC) the context states after every macroblock against the oracle (pinned on the reference's
    code by tests/golden/*subme6* / *subme7* and tests/test_oracle_vs_ref_live.py); two chained P frames"""
    import orc
    from pcamv_amd.synth import make_clip
    W, H, me, subme, qp, seed, static, cabac, psy, noise, embed = cfg[:11]
    inter = cfg[11] if len(cfg) > 11 else 0x10
    clip = make_clip(W, H, 3, seed=seed, static_cols=static, noise=noise)
    mvr = pc.level_mv_range(W, H)
    op = orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter | 1, cabac=cabac, psy_rd=psy)
    p = _params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr, cabac=cabac, psy_fix8=op.i_psy_rd, chroma_qp_offset=op.i_chroma_qp_offset)
    enc = pc.Encoder(p)
    o = orc.Oracle(op)
    ho = o.debug_state_hash()
    if cabac:
        enc.debug_state_hash(True)
    ref, prev = clip[0], (None, None)
    for t in (1, 2):
        enc.set_ref(*ref, *prev); enc.upload_fenc(*clip[t])
        o.set_ref(*ref, *prev); o.set_fenc(*clip[t])
        mbs, rec = enc.analyse_pframe(qp, embed=embed)
        mbs_o, rec_o = o.analyse_pframe(qp, embed)
        if cabac:
            bad = np.nonzero(enc.state_hash_fetch() != ho)[0]
            assert len(bad) == 0, f"frame {t}: CABAC context states differ from macroblock {bad[0]} on ({len(bad)} in all)"
        for f in mbs.dtype.names:
            assert np.array_equal(mbs[f], mbs_o[f]), f"frame {t}: {f} at MBs {np.argwhere((mbs[f] != mbs_o[f]).reshape(len(mbs), -1).any(1)).ravel()[:6]}"
        for a, b in zip(rec, rec_o):
            assert np.array_equal(a, b), f"frame {t}: reconstruction"
        prev = helpers.mv_field(mbs["mv"], W // 16, H // 16)
        ref = rec
    enc.close(); o.close()


@pytest.mark.parametrize("cfg", [(176, 144, "hex", 5, 0x10, 0.5), (176, 144, "umh", 4, 0x30, 0.5),
                                 (352, 288, "dia", 3, 0x10, 35.0), (320, 240, "hex", 1, 0x10, 0.25)])
def test_full_pipeline_matches_oracle_and_extracts(pc, cfg):
    """analysis + RCA + cover/cost assembly + STC + flips vs the oracle on seeded synthetic frames;
    then the payload is extracted from the final MVs: BER must be 0."""
    import orc
    from pcamv_amd.synth import make_clip
    W, H, me, subme, inter, rate = cfg
    clip = make_clip(W, H, 3, seed=31, static_cols=64)
    mvr = pc.level_mv_range(W, H)
    p = _params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr)
    enc = pc.Encoder(p)
    o = orc.Oracle(orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter))
    ref, prev = clip[0], (None, None)
    for t in (1, 2):
        enc.set_ref(*ref, *prev); enc.upload_fenc(*clip[t])
        o.set_ref(*ref, *prev); o.set_fenc(*clip[t])
        assert np.array_equal(enc.ref_planes(), o.ref_planes())
        mbs, rec = enc.analyse_pframe(28, embed=1)
        mbs_o, rec_o = o.analyse_pframe(28, 1)
        for f in mbs.dtype.names:
            assert np.array_equal(mbs[f], mbs_o[f]), f"frame {t}: {f}"
        for a, b in zip(rec, rec_o):
            assert np.array_equal(a, b)
        emb, emb_o = enc.embed_pframe(rate), o.embed_pframe(mbs_o, rate)
        assert (emb["n"], emb["m"], emb["stc_ok"], emb["num_flip"]) == (emb_o["n"], emb_o["m"], emb_o["stc_ok"], emb_o["num_flip"])
        for k in ("cover", "rho", "message", "stego", "flip"):
            assert np.array_equal(emb[k], emb_o[k]), f"frame {t}: {k}"
        final = enc.final_mvs(mbs)
        assert np.array_equal(final["mv"], o.final_mvs(mbs_o, emb_o)["mv"])
        lsb = helpers.carrier_lsbs(final)
        assert np.array_equal(lsb, emb["stego"])
        if emb["stc_ok"] == 1 and emb["m"] >= 10:
            assert np.array_equal(pc.stc_extract(lsb, emb["m"]), emb["message"]), "BER != 0"
        prev = helpers.mv_field(final["mv"], W // 16, H // 16)
        ref = rec
    enc.close(); o.close()


def test_embedding_edge_cases(pc):
    """user-supplied message, message longer than the cover (stc_embed fails -> all 1-bits flip,
    encoder.c:1843 ignores the return), and a frame whose every MB is skipped (n = 0)."""
    import orc
    from pcamv_amd.synth import make_clip
    W, H = 176, 144
    clip = make_clip(W, H, 2, seed=4, static_cols=48)
    # (the sub-matrix widths beyond the tables draw their columns from a generator that is process-wide in the reference and in the
    # oracle, per context in the library: a fresh context goes with the generator's initial state, whatever ran before in this process)
    orc.lib().orc_stc_lcg_reset(1)
    p = _params(pc, W, H, 1, 5, 0x10, 64, tscale=0)
    enc = pc.Encoder(p)
    o = orc.Oracle(orc.make_params(W, H, mv_range=64, tscale=0))
    enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1]); o.set_ref(*clip[0]); o.set_fenc(*clip[1])
    mbs, _ = enc.analyse_pframe(26, 1); mbs_o, _ = o.analyse_pframe(26, 1)
    msg = np.random.default_rng(0).integers(0, 2, 40).astype(np.uint8)
    a, b = enc.embed_pframe(40.0, msg), o.embed_pframe(mbs_o, 40.0, msg)
    for k in ("message", "stego", "flip"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(a["message"], msg)
    n = a["n"]
    # short messages: below 10 bits the reference's forward and backward passes shorten the last columns differently
    # (embed.h:462 vs 523) -- reproduced, not fixed; few bits over many MVs: sub-matrix widths beyond the 20 tabulated
    # ones (random columns from the LCG, up to 256 wide, embed.h:286)
    for bits in (2, 3, 5, 7, 9, 10, 11):
        a, b = enc.embed_pframe(bits + 0.5), o.embed_pframe(mbs_o, bits + 0.5)
        assert a["m"] == bits and b["m"] == bits and a["stc_ok"] == b["stc_ok"], bits
        for k in ("message", "stego", "flip"):
            assert np.array_equal(a[k], b[k]), (bits, k)
    a, b = enc.embed_pframe(float(n + 5)), o.embed_pframe(mbs_o, float(n + 5))
    assert a["stc_ok"] == 0 and b["stc_ok"] == 0 and np.array_equal(a["flip"], a["cover"].astype(np.int8)) and np.array_equal(a["flip"], b["flip"])
    # a bits-per-frame rate beyond the arrays' capacity (16 per macroblock): fails like any m > n, writes nothing out of bounds,
    # and the rand() stream has advanced by m all the same (the next frame's message continues from there)
    cap = 16 * len(mbs)
    a = enc.embed_pframe(float(cap + 2000))
    assert a["stc_ok"] == 0 and a["m"] == cap + 2000 and len(a["message"]) == cap and np.array_equal(a["flip"], a["cover"].astype(np.int8))
    nxt = enc.embed_pframe(40.5)
    consumed = sum(bits for bits in (2, 3, 5, 7, 9, 10, 11)) + (n + 5)        # rand() calls before the oversized request (the first embedding had a caller's message)
    stream = np.array([v & 1 for v in orc.glibc_rand(consumed + cap + 2000 + 40)], np.uint8)
    assert np.array_equal(nxt["message"], stream[consumed + cap + 2000:])
    # identical frames, no noise: every MB becomes P_SKIP, no carriers
    enc.set_ref(*clip[0]); enc.upload_fenc(*clip[0])
    mbs, _ = enc.analyse_pframe(30, 1)
    assert (mbs["i_type"] == pc.P_SKIP).all()
    e = enc.embed_pframe(0.5)
    assert e["n"] == 0 and e["m"] == 0 and e["num_flip"] == 0
    enc.close(); o.close()


def test_stc_forward_with_four_states_per_thread(pc, monkeypatch):
    """the forward Viterbi of the embedding stage runs 4 trellis states per thread (4 waves per frame) from 1024 GOPs in flight on, 2
    below -- more than a test can afford, so the instance is forced here: the same cover / rho / stego / flip vectors and payloads,
    tabulated sub-matrix widths and generated ones, short messages"""
    monkeypatch.setenv("PCAMV_STC_STATES", "4")
    test_full_pipeline_matches_oracle_and_extracts(pc, (176, 144, "hex", 5, 0x10, 0.5))
    test_full_pipeline_matches_oracle_and_extracts(pc, (352, 288, "dia", 3, 0x10, 35.0))
    test_embedding_edge_cases(pc)


def test_open_rejects_unsupported(pc):
    p = pc.param_default(176, 144)
    pc.param_parse(p, "subme", 8)            # RD refinement of the MVs: not built (and disabled in the fork's P frames, analyse.c:3112)
    with pytest.raises(pc.PcamvError):
        pc.Encoder(p)
    p = pc.param_default(176, 144)
    pc.param_parse(p, "partitions", "all")   # subme 6 (the default) with sub-8x8 partitions: x264_rd_cost_part, built in round 3
    pc.Encoder(p).close()
    p = pc.param_default(176, 144)
    pc.param_parse(p, "me", "tesa")          # the default subme 6 with --me tesa: built in round 3
    pc.Encoder(p).close()
    p = pc.param_default(170, 144)
    with pytest.raises(pc.PcamvError):
        pc.Encoder(p)
    p = pc.param_default(176, 144)      # --me tesa keeps its survivor list in LDS: up to me_range 16
    pc.param_parse(p, "me", "tesa"); pc.param_parse(p, "subme", 5)
    p.i_me_range = 24
    with pytest.raises(pc.PcamvError):
        pc.Encoder(p)


def test_schedules_agree(pc):
    """the persistent dataflow launch and the per-anti-diagonal launches are the same computation"""
    import os
    from pcamv_amd.synth import make_clip
    W, H = 352, 288
    clip = make_clip(W, H, 2, seed=5, static_cols=32)
    out = {}
    for sched in ("diag", "flow"):
        if sched == "diag":
            os.environ["PCAMV_SCHED"] = "diag"
        try:
            enc = pc.Encoder(_params(pc, W, H, pc.ME_NAMES["umh"], 5, 0x10, pc.level_mv_range(W, H)))
        finally:
            os.environ.pop("PCAMV_SCHED", None)
        enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1])
        mbs, rec = enc.analyse_pframe(27, embed=1)
        emb = enc.embed_pframe(0.5)
        out[sched] = (mbs, rec, emb)
        enc.close()
    a, b = out["diag"], out["flow"]
    for f in a[0].dtype.names:
        assert np.array_equal(a[0][f], b[0][f]), f
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x, y)
    for k in ("cover", "rho", "message", "stego", "flip"):
        assert np.array_equal(a[2][k], b[2][k]), k


def test_closed_loop_schedules_agree_at_scale(pc):
    """1080p, 32 closed GOPs (four per XCD queue), two closed-loop steps: the dataflow kernels (neighbour hand-off by
    write-through stores + agent-scope loads, work stealing between the queues) against the per-anti-diagonal
    launches, whose ordering comes from kernel boundaries -- records, flip maps and deblocked pictures of every GOP.
    A size-independent property: the two schedules are the same computation."""
    import os
    import torch
    from pcamv_amd.synth import make_clip
    W, H, G, qp = 1920, 1088, 32, 26
    clip = make_clip(W, H, 4, seed=13)
    dev = torch.device("cuda", 0)
    d = [[torch.from_numpy(np.ascontiguousarray(pl)).to(dev) for pl in fr] for fr in clip]
    p = _params(pc, W, H, pc.ME_NAMES["umh"], 5, 0x10, pc.level_mv_range(W, H))
    out = {}
    for sched in ("diag", "flow"):
        if sched == "diag":
            os.environ["PCAMV_SCHED"] = "diag"
        try:
            encs = [pc.Encoder(p) for _ in range(G)]
            batch = pc.Batch(encs)
        finally:
            os.environ.pop("PCAMV_SCHED", None)
        batch.set_closed_loop(True)
        res = []
        for t in (0, 1):
            for g, enc in enumerate(encs):
                if t == 0:
                    a = d[g % 3]
                    enc.set_ref_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), enc.PREV_INTERNAL, enc.PREV_INTERNAL)
                else:
                    r = enc.recon_device()
                    enc.set_ref_device(r[0], r[1], r[2], enc.PREV_INTERNAL, enc.PREV_INTERNAL)
                b = d[(g + t) % 3 + 1]
                enc.set_fenc_device(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
            batch.step(qp, 0.5, 0)
            for enc in encs:
                mbs, emb = enc.fetch_results(want_embed=True)
                res.append((mbs, emb["flip"].copy(), [x.copy() for x in enc.fetch_recon()]))
        out[sched] = res
        batch.close()
        for enc in encs:
            enc.close()
    for i, (a, b) in enumerate(zip(out["diag"], out["flow"])):
        for f in a[0].dtype.names:
            assert np.array_equal(a[0][f], b[0][f]), f"step {i // G} GOP {i % G}: {f}"
        assert np.array_equal(a[1], b[1]), f"step {i // G} GOP {i % G}: flips"
        for x, y, nm in zip(a[2], b[2], "yuv"):
            assert np.array_equal(x, y), f"step {i // G} GOP {i % G}: deblocked {nm}"


def test_1080p_batch_step_matches_oracle(pc):
    """BASELINE's size through the batch API (device-resident planes, several closed GOPs advanced by one
    dataflow launch): every GOP's record, embedding vectors and extracted payload against the oracle."""
    import torch
    import orc
    from pcamv_amd.synth import make_clip
    W, H = 1920, 1088
    clip = make_clip(W, H, 3, seed=13)
    dev = torch.device("cuda", 0)
    d = [[torch.from_numpy(np.ascontiguousarray(pl)).to(dev) for pl in fr] for fr in clip]
    mvr = pc.level_mv_range(W, H)
    p = _params(pc, W, H, pc.ME_NAMES["umh"], 5, 0x10, mvr)
    encs = [pc.Encoder(p) for _ in range(2)]
    batch = pc.Batch(encs)
    for g, enc in enumerate(encs):       # GOP g: reference = frame g, source = frame g + 1
        enc.set_ref_device(d[g][0].data_ptr(), d[g][1].data_ptr(), d[g][2].data_ptr(), None, None)
        enc.set_fenc_device(d[g + 1][0].data_ptr(), d[g + 1][1].data_ptr(), d[g + 1][2].data_ptr())
    batch.step(26, 0.5, 0)
    for g, enc in enumerate(encs):
        mbs, emb = enc.fetch_results(want_embed=True)
        o = orc.Oracle(orc.make_params(W, H, me="umh", subme=5, mv_range=mvr))   # own message stream, like each context
        o.set_ref(*clip[g]); o.set_fenc(*clip[g + 1])
        mbs_o, _ = o.analyse_pframe(26, 1)
        for f in mbs.dtype.names:
            assert np.array_equal(mbs[f], mbs_o[f]), f"GOP {g}: {f}"
        emb_o = o.embed_pframe(mbs_o, 0.5)
        assert (emb["n"], emb["m"], emb["stc_ok"], emb["num_flip"]) == (emb_o["n"], emb_o["m"], emb_o["stc_ok"], emb_o["num_flip"])
        for k in ("cover", "rho", "message", "stego", "flip"):
            assert np.array_equal(emb[k], emb_o[k]), f"GOP {g}: {k}"
        final = enc.final_mvs(mbs)
        assert np.array_equal(pc.stc_extract(helpers.carrier_lsbs(final), emb["m"]), emb["message"]), "BER != 0"
        o.close()
    batch.close()
    for enc in encs:
        enc.close()


SWEEP = [
    # (W, H, me, me_range, subme, inter, qp, chroma_me, fast_pskip, dct_decimate, chroma_qp_offset, seed, static_cols)
    (176, 144, "dia", 16, 1, 0x10, 12, 1, 1, 1, 0, 41, 48),      # qp < 24: rounding dequantiser
    (176, 144, "hex", 16, 3, 0x10, 20, 1, 1, 1, 3, 42, 32),      # chroma QP offset
    (176, 144, "umh", 24, 5, 0x10, 35, 0, 1, 1, 0, 43, 0),       # no chroma ME, wide UMH range
    (176, 144, "umh", 8, 4, 0x30, 45, 1, 0, 1, -2, 44, 64),      # p4x4, no fast skip, high qp
    (176, 144, "hex", 16, 5, 0x30, 30, 1, 1, 0, 0, 45, 16),      # no decimation
    (176, 144, "esa", 8, 2, 0x10, 28, 1, 1, 1, 0, 46, 48),       # exhaustive, subme 2
    (320, 240, "umh", 16, 5, 0x10, 24, 1, 1, 1, 0, 47, 96),      # borders of a wider picture, many skips
    (176, 144, "esa", 16, 5, 0x30, 33, 1, 1, 1, 0, 48, 32),      # exhaustive with p4x4
    (1280, 720, "hex", 16, 5, 0x10, 26, 1, 1, 1, 0, 11, 320),    # BASELINE config 2's size and search (720p, --me hex)
    (640, 480, "esa", 16, 5, 0x10, 26, 1, 1, 1, 0, 17, 128),     # config 5's search (--me esa) on a larger picture
    (176, 144, "tesa", 16, 4, 0x10, 30, 1, 1, 1, 0, 49, 32),     # Hadamard exhaustive search: SATD full-pel metric, ADS / SAD thresholds
    (176, 144, "tesa", 8, 1, 0x30, 24, 1, 1, 1, 0, 50, 0),       # subme 1: the same search with the SAD metric, p4x4
    (320, 240, "tesa", 16, 5, 0x30, 28, 1, 1, 1, 0, 51, 64),     # wider picture: windows clipped by the MV limits
]


@pytest.mark.parametrize("cfg", SWEEP, ids=[f"{c[2]}_r{c[3]}_s{c[4]}_i{c[5]:x}_qp{c[6]}" for c in SWEEP])
def test_option_sweep_matches_oracle(pc, cfg):
    """options and quantiser ranges the fixtures do not reach (the oracle is pinned to the reference on the
    fixtures; here it carries that pin to more of the option space): two chained P frames, record +
    reconstruction + RCA costs bit-exact"""
    import orc
    from pcamv_amd.synth import make_clip
    W, H, me, me_range, subme, inter, qp, cme, fps, dec, cqo, seed, static = cfg
    clip = make_clip(W, H, 3, seed=seed, static_cols=static)
    mvr = pc.level_mv_range(W, H)
    p = _params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr, me_range)
    p.b_chroma_me, p.b_fast_pskip, p.b_dct_decimate, p.i_chroma_qp_offset = cme, fps, dec, cqo
    enc = pc.Encoder(p)
    o = orc.Oracle(orc.make_params(W, H, me=me, me_range=me_range, subme=subme, mv_range=mvr, chroma_me=cme, fast_pskip=fps,
                                   dct_decimate=dec, inter=inter, chroma_qp_offset=cqo))
    ref, prev = clip[0], (None, None)
    for t in (1, 2):
        enc.set_ref(*ref, *prev); enc.upload_fenc(*clip[t])
        o.set_ref(*ref, *prev); o.set_fenc(*clip[t])
        mbs, rec = enc.analyse_pframe(qp, embed=1)
        mbs_o, rec_o = o.analyse_pframe(qp, 1)
        for f in mbs.dtype.names:
            assert np.array_equal(mbs[f], mbs_o[f]), f"frame {t}: {f} at MBs {np.argwhere((mbs[f] != mbs_o[f]).reshape(len(mbs), -1).any(1)).ravel()[:6]}"
        for a, b in zip(rec, rec_o):
            assert np.array_equal(a, b), f"frame {t}: reconstruction"
        prev = helpers.mv_field(mbs["mv"], W // 16, H // 16)
        ref = rec
    enc.close(); o.close()


PASS2_GPU = [
    # (W, H, me, subme, inter, qp, seed, static_cols, flip_rate)
    (176, 144, "hex", 5, 0x00, 26, 5, 48, 0.3),
    (352, 288, "umh", 4, 0x10, 22, 7, 0, 0.5),        # partitions, no skips
    (352, 288, "hex", 5, 0x00, 38, 5, 0, 0.4),        # moving skips re-predicted from flipped neighbours
    (320, 240, "hex", 5, 0x30, 18, 9, 96, 0.25),      # p4x4 carriers, fine quantiser (many bS = 2 edges)
    (176, 144, "dia", 3, 0x10, 44, 8, 0, 0.35),       # coarse quantiser: strong filtering
    (176, 144, "hex", 5, 0x10, 14, 11, 32, 0.3),      # qp <= 15: only macroblock edges are filtered
]


@pytest.mark.parametrize("cfg", PASS2_GPU, ids=[f"{c[0]}x{c[1]}_{c[2]}_i{c[4]:x}_qp{c[5]}" for c in PASS2_GPU])
def test_pass2_and_loop_filter_match_oracle(pc, cfg):
    """final MVs (flips applied, skips re-predicted), pass-2 reconstruction and the deblocked picture against the
    oracle (itself pinned on the reference's second pass + x264_frame_deblock_row by the CPU suite); once with an
    explicit flip map, once with the device flip map of the embedding stage"""
    import orc
    from pcamv_amd.synth import make_clip
    W, H, me, subme, inter, qp, seed, static, rate = cfg
    clip = make_clip(W, H, 2, seed=seed, static_cols=static)
    mvr = pc.level_mv_range(W, H)
    enc = pc.Encoder(_params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr))
    o = orc.Oracle(orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter))
    enc.set_ref(*clip[0]); enc.upload_fenc(*clip[1])
    o.set_ref(*clip[0]); o.set_fenc(*clip[1])
    mbs, _ = enc.analyse_pframe(qp, embed=1)
    mbs_o, _ = o.analyse_pframe(qp, 1)
    n = len(helpers.carrier_lsbs(mbs))
    flips = (np.random.default_rng(seed).random(n) < rate).astype(np.uint8)
    fin, rec, dbk = enc.pass2_pframe(flips)
    fo, nnz_o, rec_o, dbk_o, k = o.pass2_pframe(qp, mbs_o, flips)
    assert k == n
    assert np.array_equal(fin["mv"], fo["mv"]), np.argwhere((fin["mv"] != fo["mv"]).reshape(len(fin), -1).any(1)).ravel()[:8]
    for a, b, nm in zip(rec, rec_o, "yuv"):
        assert np.array_equal(a, b), f"pass-2 reconstruction {nm}"
    for a, b, nm in zip(dbk, dbk_o, "yuv"):
        assert np.array_equal(a, b), f"deblocked {nm}: {np.argwhere(a != b)[:6].tolist()}"
    assert qp < 20 or any((a != b).any() for a, b in zip(rec, dbk))        # (alpha = 0 below indexA 16: nothing to filter)
    # the same through the embedding stage's own flip map
    mbs, _ = enc.analyse_pframe(qp, embed=1)
    emb = enc.embed_pframe(0.5)
    fin2, _, dbk2 = enc.pass2_pframe()
    o2 = orc.Oracle(orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter))
    o2.set_ref(*clip[0]); o2.set_fenc(*clip[1])
    mbs_o2, _ = o2.analyse_pframe(qp, 1)
    emb_o = o2.embed_pframe(mbs_o2, 0.5)
    assert np.array_equal(emb["flip"], emb_o["flip"])
    fo2, _, _, dbk_o2, _ = o2.pass2_pframe(qp, mbs_o2, (np.asarray(emb_o["flip"]) == 1).astype(np.uint8))
    assert np.array_equal(fin2["mv"], fo2["mv"])
    for a, b in zip(dbk2, dbk_o2):
        assert np.array_equal(a, b)
    # ... and a second pass over the SAME analysis once more, with another map: by now the frame's reconstruction planes hold the
    # filtered picture, and nothing of it may be taken for the first pass' pixels (the library reuses those only while they are there)
    flips3 = (np.random.default_rng(seed + 100).random(n) < rate).astype(np.uint8)
    fin3, rec3, dbk3 = enc.pass2_pframe(flips3)
    fo3, _, rec_o3, dbk_o3, _ = o2.pass2_pframe(qp, mbs_o2, flips3)
    assert np.array_equal(fin3["mv"], fo3["mv"])
    for a, b, nm in zip(rec3, rec_o3, "yuv"):
        assert np.array_equal(a, b), f"second call, pass-2 reconstruction {nm}"
    for a, b, nm in zip(dbk3, dbk_o3, "yuv"):
        assert np.array_equal(a, b), f"second call, deblocked {nm}"
    enc.close(); o.close(); o2.close()


def _closed_loop_vs_oracle(pc, W, H, me, subme, qp, n_gops, steps, seed0, emrate=0.5, statics=(0, 64, 128), noise=6, hashes=False, inter=0x10):
    """GOPs advanced together through closed-loop steps (dataflow analysis, embedding, then pass 2 + loop filter through the same
    dataflow queue; every later step's reference is the step's own deblocked picture and final motion field, both taken from the
    device): records, embedding vectors, deblocked pictures vs the oracle, and the payload back out of the final motion vectors
    (the extractor carries the STC column generator's state from frame to frame, one process per GOP)."""
    import torch
    import orc
    from pcamv_amd.synth import make_clip
    clips = [make_clip(W, H, steps + 1, seed=seed0 + g, static_cols=statics[g % len(statics)], noise=noise) for g in range(n_gops)]
    orc.lib().orc_stc_lcg_reset(1)          # fresh contexts: the column generator's initial state on both sides (it is process-wide in the oracle)
    dev = torch.device("cuda", 0)
    d = [[[torch.from_numpy(np.ascontiguousarray(pl)).to(dev) for pl in fr] for fr in clip] for clip in clips]
    mvr = pc.level_mv_range(W, H)
    rd = subme >= 6
    op = orc.make_params(W, H, me=me, subme=subme, mv_range=mvr, inter=inter | 1 if rd else inter)
    p = _params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr, psy_fix8=op.i_psy_rd, chroma_qp_offset=op.i_chroma_qp_offset) if rd \
        else _params(pc, W, H, pc.ME_NAMES[me], subme, inter, mvr)
    encs = [pc.Encoder(p) for _ in range(n_gops)]
    batch = pc.Batch(encs)
    batch.set_closed_loop(True)
    oracles = [orc.Oracle(op) for _ in range(n_gops)]
    ohash = [o.debug_state_hash() for o in oracles] if hashes else None      # (CABAC: the 460 context states after every macroblock)
    if hashes:
        for enc in encs:
            enc.debug_state_hash(True)
    lcgs = [pc.StcLcg(1) for _ in range(n_gops)]
    refs = [clips[g][0] for g in range(n_gops)]
    prevs = [(None, None)] * n_gops
    bits = 0
    for t in range(1, steps + 1):
        for g, enc in enumerate(encs):
            if t == 1:      # (the internal field of a fresh context holds no motion: same as no previous frame)
                enc.set_ref_device(d[g][0][0].data_ptr(), d[g][0][1].data_ptr(), d[g][0][2].data_ptr(), enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            else:
                r = enc.recon_device()
                enc.set_ref_device(r[0], r[1], r[2], enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            enc.set_fenc_device(d[g][t][0].data_ptr(), d[g][t][1].data_ptr(), d[g][t][2].data_ptr())
        batch.step(qp, emrate, 0)
        for g, enc in enumerate(encs):
            o = oracles[g]
            mbs, emb = enc.fetch_results(want_embed=True)
            o.set_ref(*refs[g], *prevs[g]); o.set_fenc(*clips[g][t])
            mbs_o, _ = o.analyse_pframe(qp, 1)
            if hashes:
                bad = np.nonzero(enc.state_hash_fetch() != ohash[g])[0]
                assert len(bad) == 0, f"step {t} GOP {g}: CABAC context states differ from macroblock {bad[0]} on ({len(bad)} in all)"
            for f in mbs.dtype.names:
                assert np.array_equal(mbs[f], mbs_o[f]), f"step {t} GOP {g}: {f}"
            emb_o = o.embed_pframe(mbs_o, emrate)
            assert (emb["n"], emb["m"], emb["stc_ok"], emb["num_flip"]) == (emb_o["n"], emb_o["m"], emb_o["stc_ok"], emb_o["num_flip"])
            for k in ("cover", "rho", "message", "stego", "flip"):
                assert np.array_equal(emb[k], emb_o[k]), f"step {t} GOP {g}: {k}"
            fo, _, _, dbk_o, _ = o.pass2_pframe(qp, mbs_o, (np.asarray(emb_o["flip"]) == 1).astype(np.uint8))
            dbk = enc.fetch_recon()
            for a, b, nm in zip(dbk, dbk_o, "yuv"):
                assert np.array_equal(a, b), f"step {t} GOP {g}: deblocked {nm}"
            if emb["m"] > 0:
                assert emb["stc_ok"] == 1
                final = enc.final_mvs(mbs)
                assert np.array_equal(pc.stc_extract(helpers.carrier_lsbs(final), emb["m"], lcg=lcgs[g]), emb["message"]), f"step {t} GOP {g}: BER != 0"
                bits += emb["m"]
            refs[g] = dbk_o
            prevs[g] = helpers.mv_field(fo["mv"], W // 16, H // 16)
    batch.close()
    for enc in encs:
        enc.close()
    for o in oracles:
        o.close()
    return bits


@pytest.mark.parametrize("n_gops", [2, 16])
def test_closed_loop_batch_step_matches_oracle(pc, n_gops):
    """16 GOPs = one queue per XCD, hand-offs inside and across XCDs (2 GOPs share a single queue)."""
    assert _closed_loop_vs_oracle(pc, 352, 288, "hex", 5, 30, n_gops, 2, 61) > 0


# BASELINE.json's configurations at their own sizes and options (synthetic clips of the SURVEY 8(d) generator), closed loop,
# embedding on, payload extracted again
def test_config1_cif_dia_35_bits_per_frame(pc):
    """config 1: CIF, --me dia, the reference's default --subme (6), "1 kbit" = --emrate 35 (bits per P frame): few bits over many
    carriers = sub-matrix widths beyond the tables, columns from the LCG, which the extractor follows over the frames"""
    assert _closed_loop_vs_oracle(pc, 352, 288, "dia", 6, 26, 1, 4, 7, emrate=35.0, statics=(0,), noise=10) == 4 * 35


def test_config2_720p_hex(pc):
    assert _closed_loop_vs_oracle(pc, 1280, 720, "hex", 6, 26, 1, 2, 11, statics=(320,)) > 1000


def test_config3_1080p_umh_subme7_closed_loop(pc):
    """config 3 as written (the bench workload): two GOPs, two chained steps"""
    assert _closed_loop_vs_oracle(pc, 1920, 1088, "umh", 7, 26, 2, 2, 13, statics=(480, 0)) > 4000


def test_config3_1080p_second_pass_in_runs_of_eight(pc, monkeypatch):
    """config 3 with the second pass as the bench runs it: tasks of eight macroblocks of a row, each one tile in LDS (P2Unit) -- at 1080p's
    own geometry (rows of 120 macroblocks = 15 runs), which the small-picture tests of the run sizes do not have"""
    monkeypatch.setenv("PCAMV_PASS2_UNIT", "8")
    assert _closed_loop_vs_oracle(pc, 1920, 1088, "umh", 7, 26, 2, 2, 19, statics=(0, 480)) > 4000


def test_config5_2160p_esa(pc):
    """config 5: 3840x2160, exhaustive search, one GOP, one full P frame (32400 macroblocks in one chain)"""
    assert _closed_loop_vs_oracle(pc, 3840, 2160, "esa", 6, 26, 1, 1, 17, statics=(1920,)) > 4000


@pytest.mark.parametrize("inst", ["hi", "lo", "spec", "spec2", "spec4"])
def test_rd_instances_agree(pc, monkeypatch, inst):
    """the RD instance has five builds (pcamv_rd*.hip): the ones that hand a chain on speculatively after the 16x16 search at 1 (the
    default of every small test here), 2 and 4 waves per SIMD, chosen by the number of chains, and the plain ones at 1 and 4 waves
    per SIMD (no chain to speculate on -- CAVLC --, or thousands of chains); here each is forced onto the same small batches: CABAC (one chain per frame) closed loop, then the
    CAVLC sizes (wavefront order: no chain to speculate on, "spec" falls back) and a noisy CABAC case through the sweep"""
    monkeypatch.setenv("PCAMV_RD_INSTANCE", inst)
    assert _closed_loop_vs_oracle(pc, 352, 288, "umh", 7, 26, 3, 2, 71, hashes=True) > 0
    test_rd_mode_decision_matches_oracle(pc, RD_SWEEP[1])
    test_rd_mode_decision_matches_oracle(pc, RD_SWEEP[6])
    test_rd_mode_decision_matches_oracle(pc, RD_SWEEP[0])


FINAL_SLICES = ["pslice_qcif_hex_subme5_final", "pslice_cif_umh_subme7_final", "pslice_cavlc_cif_umh_subme7_final"]


@pytest.mark.parametrize("name", FINAL_SLICES)
def test_final_mvs_are_what_a_decoder_reads(pc, name):
    """SURVEY 8f rank 1 chained to the GPU: pass 1 + embedding + pass 2 on the GPU from the fixture's pictures; the slice the
    REFERENCE's entropy coder wrote for the same frame (its own second pass with the same flips) is parsed by the MV-syntax
    extractor: types, partitions and final motion vectors agree, and the payload the GPU embedded comes back out of the parsed
    stream's motion (decode-side BER = 0 on GPU output)."""
    g = helpers.load(name)
    W, H, qp = int(g["width"]), int(g["height"]), int(g["qp"])
    cabac = int(g["cabac"])
    p = _params(pc, W, H, int(g["me"]), int(g["subme"]), int(g["inter"]) & 0x30, int(g["mv_range"]), cabac=cabac)
    enc = pc.Encoder(p)
    enc.set_ref(g["ref_y"], g["ref_u"], g["ref_v"]); enc.upload_fenc(g["fenc_y"], g["fenc_u"], g["fenc_v"])
    mbs, _ = enc.analyse_pframe(qp, embed=1)
    emb = enc.embed_pframe(0.5)
    fin, _, _ = enc.pass2_pframe()
    rbsp, _, _ = pc.nal_to_rbsp(g["nal"].tobytes())
    got = pc.parse_pslice_at(rbsp, int(g["nal_hdr_bits"]), W // 16, H // 16, qp if cabac else None)
    assert np.array_equal(got["i_type"], mbs["i_type"]) and np.array_equal(got["i_partition"], mbs["i_partition"])
    bad = np.argwhere((got["mv"] != fin["mv"]).reshape(len(fin), -1).any(1)).ravel()
    assert len(bad) == 0, f"final motion differs from the stream's at macroblocks {bad[:8].tolist()}"
    got["used"] = mbs["used"]
    lsb = helpers.carrier_lsbs(got)
    assert len(lsb) == emb["n"] == int(g["n"]) and np.array_equal(lsb, emb["stego"])
    msg = pc.stc_extract(lsb, emb["m"])
    assert np.array_equal(msg, emb["message"]) and np.array_equal(msg, g["message"]), "decode-side BER != 0"
    enc.close()


def test_mvsyntax_extractor_rows_on_the_gpu_box(pc):
    """the extractor's own suite (host code of the same library; tests/test_mvsyntax.py, CPU suite) once more where the GPU tests
    run, so that the row is covered by the run that records which native code was loaded"""
    import test_mvsyntax as tm
    for name in tm.FIXTURES:
        tm.test_parser_reads_back_what_the_reference_coded(name)
        tm.test_slice_data_inside_a_nal_unit_behind_a_header(name)
        if name.endswith("_final"):
            tm.test_payload_comes_back_out_of_the_stream(name)
    tm.test_damaged_streams_are_reported()
    tm.test_skip_run_beyond_the_picture_is_reported()


def test_batch_survives_a_closed_context(pc):
    """a context closed before its batch: every batch entry point reports it (PCAMV_EINVAL), none dereferences the dead slot"""
    import ctypes as C
    W, H = 176, 144
    encs = [pc.Encoder(_params(pc, W, H, 1, 5, 0x10, 64)) for _ in range(3)]
    batch = pc.Batch(encs)
    encs[1].close()
    assert batch.dominant_kernel() == "k_analyse_flow"
    batch.kernel_time(reset=True)
    with pytest.raises(pc.PcamvError, match="closed"):
        batch.step(26, 0.5, 0)
    buf = np.zeros(3 * 99 * 236, np.uint8)
    with pytest.raises(pc.PcamvError, match="closed"):
        batch.copy_results_async(buf.ctypes.data, 99 * 236)
    batch.close()
    encs[0].close(); encs[2].close()


@pytest.mark.parametrize("inst", ["hi", "spec4"])
def test_bench_regime_matches_oracle(pc, monkeypatch, inst):
    """The regime bench.py times, against the oracle: --me umh --subme 7 with CABAC (analyse.c:2117-2186, rdo.c:139-171), the
    4-waves-per-SIMD build of the RD kernel ("hi": the default run's 4096 chains; "spec4": the speculative one its g_sweep / clip_600
    blocks run at 705..3584 chains), eight per-XCD queues (40 closed GOPs = five chains per queue), fewer waves than chains
    (12: a wave that finishes a macroblock takes whatever chain's next one is ready, queues without a wave of their own are drained
    by work stealing), the second pass in tasks of 8 macroblocks (one LDS tile each), the embedding stage's forward pass at 4 trellis states per thread; three closed-loop steps; of EVERY GOP the records, the context states
    after every macroblock, the embedding vectors, the deblocked planes and the payload back out of the final motion vectors."""
    monkeypatch.setenv("PCAMV_RD_INSTANCE", inst)
    monkeypatch.setenv("PCAMV_PASS2_UNIT", "8")
    monkeypatch.setenv("PCAMV_FLOW_WAVES", "12")
    monkeypatch.setenv("PCAMV_STC_STATES", "4")          # the forward Viterbi as thousands of frames in flight run it
    assert _closed_loop_vs_oracle(pc, 352, 288, "umh", 7, 26, 40, 3, 171, hashes=True) > 0


@pytest.mark.parametrize("waves", [1, 2, 5])
def test_speculative_chain_with_few_waves(pc, monkeypatch, waves):
    """the speculative raster chain when waves are scarcer than the work it exposes (1 wave: every macroblock finds its predecessor
    final; 2 / 5 waves over 3 chains: hand-offs wait for older macroblocks held by other waves), noisy pictures (partitions: many
    macroblocks do NOT end as the 16x16 they announced, their successors start over); closed loop, context states of every macroblock"""
    monkeypatch.setenv("PCAMV_FLOW_SPEC", "1")
    monkeypatch.setenv("PCAMV_FLOW_WAVES", str(waves))
    assert _closed_loop_vs_oracle(pc, 176, 144, "hex", 6, 24, 3, 2, 91, statics=(0, 32, 0), noise=30, hashes=True) > 0


@pytest.mark.parametrize("unit", [3, 8])
def test_pass2_task_sizes_match_oracle(pc, monkeypatch, unit):
    """the second pass takes `unit` macroblocks of a row per queue task (8 by default from 256 GOPs in flight on -- more than a
    test can afford, so the size is forced here): rows of 22 macroblocks = tasks of 8 + 8 + 6 and 3 x 7 + 1, both RD levels"""
    monkeypatch.setenv("PCAMV_PASS2_UNIT", str(unit))
    assert _closed_loop_vs_oracle(pc, 352, 288, "hex", 5, 30, 3, 2, 81) > 0
    assert _closed_loop_vs_oracle(pc, 352, 288, "umh", 7, 26, 2, 2, 83) > 0
