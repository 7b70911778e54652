"""ctypes binding of oracle/libpcamv_oracle.so (TEST INFRASTRUCTURE ONLY: the CPU restatement).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcamv_oracle.so")

ME = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}
PIXEL = {"16x16": 0, "16x8": 1, "8x16": 2, "8x8": 3, "8x4": 4, "4x8": 5, "4x4": 6}
P_L0, P_8x8, P_SKIP = 4, 5, 6
I4x4, PSUB16x16, PSUB8x8 = 0x01, 0x10, 0x20


class Params(C.Structure):
    _fields_ = [("i_width", C.c_int32), ("i_height", C.c_int32), ("i_me_method", C.c_int32),
                ("i_me_range", C.c_int32), ("i_subpel_refine", C.c_int32), ("i_mv_range", C.c_int32),
                ("b_chroma_me", C.c_int32), ("b_fast_pskip", C.c_int32), ("b_dct_decimate", C.c_int32),
                ("b_cabac", C.c_int32), ("inter", C.c_uint32), ("i_chroma_qp_offset", C.c_int32),
                ("i_luma_deadzone", C.c_int32 * 2), ("i_tscale", C.c_int32), ("i_psy_rd", C.c_int32)]


class Embed(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("stc_ok", C.c_int32), ("num_flip", C.c_int32),
                ("cover", C.c_void_p), ("rho", C.c_void_p), ("message", C.c_void_p),
                ("stego", C.c_void_p), ("flip", C.c_void_p)]


MB_DTYPE = np.dtype([("i_type", "<i4"), ("i_partition", "<i4"), ("i_qp", "<i4"),
                     ("i_sub_partition", "u1", (4,)), ("ref", "i1", (16,)),
                     ("mv", "<i2", (16, 2)), ("mv_stego", "<i2", (16, 2)),
                     ("inter_stego_cost", "<i4", (16,)), ("pskip_mv", "<i2", (2,)),
                     ("mvr16", "<i2", (2,)), ("used", "u1"), ("pad", "u1", (3,))])


def make_params(width, height, me="hex", me_range=16, subme=5, mv_range=None, chroma_me=1,
                fast_pskip=1, dct_decimate=1, cabac=1, inter=PSUB16x16, chroma_qp_offset=None,
                tscale=256, psy_rd=1.0):
    """psy_rd / chroma_qp_offset as x264_validate_parameters leaves them (encoder.c:511-522): psy-RD acts from subme 6 on
    and lowers the chroma QP offset by 2 (1 below strength 0.25)"""
    if mv_range is None:
        mv_range = level_mv_range(width, height)
    psy = int(psy_rd * 256 + 0.5) if subme >= 6 else 0
    if chroma_qp_offset is None:
        chroma_qp_offset = 0 if not psy else (-1 if psy_rd < 0.25 else -2)
    p = Params(width, height, ME[me] if isinstance(me, str) else me, me_range, subme, mv_range,
               chroma_me, fast_pskip, dct_decimate, cabac, inter, chroma_qp_offset)
    p.i_luma_deadzone[0] = 21
    p.i_luma_deadzone[1] = 11
    p.i_tscale = tscale
    p.i_psy_rd = psy
    return p


def level_mv_range(width, height, fps=25):
    """x264_validate_levels / encoder.c:540-559 for 1 reference frame: lowest H.264 level whose
    frame size, MB rate and DPB admit the stream; mv_range of that level (Table A-1)."""
    mbs = (width // 16) * (height // 16)
    levels = [(10, 1485, 99, 148500, 64), (11, 3000, 396, 337500, 128), (12, 6000, 396, 891000, 128),
              (13, 11880, 396, 891000, 128), (20, 11880, 396, 891000, 128), (21, 19800, 792, 1782000, 256),
              (22, 20250, 1620, 3037500, 256), (30, 40500, 1620, 3037500, 256), (31, 108000, 3600, 6750000, 512),
              (32, 216000, 5120, 7680000, 512), (40, 245760, 8192, 12288000, 512), (41, 245760, 8192, 12288000, 512),
              (42, 522240, 8704, 13056000, 512), (50, 589824, 22080, 41400000, 512), (51, 983040, 36864, 69120000, 512)]
    for _, mbps, fs, dpb, mvr in levels:
        if fs >= mbs and mbps >= mbs * fps and dpb >= 384 * mbs * 1:
            return mvr
    return 512


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_open.restype = C.c_void_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, params):
        self.p = params
        self.w, self.h = params.i_width, params.i_height
        self.n_mb = (self.w // 16) * (self.h // 16)
        self.ctx = C.c_void_p(lib().orc_open(C.byref(params)))
        if not self.ctx:
            raise RuntimeError("orc_open failed")

    def close(self):
        if self.ctx:
            lib().orc_close(self.ctx)
            self.ctx = None

    def set_fenc(self, y, u, v):
        y, u, v = [np.ascontiguousarray(a, np.uint8) for a in (y, u, v)]
        lib().orc_set_fenc(self.ctx, _p(y), _p(u), _p(v))

    def set_ref(self, y, u, v, prev_mv=None, prev_ref=None):
        y, u, v = [np.ascontiguousarray(a, np.uint8) for a in (y, u, v)]
        if prev_mv is not None:
            prev_mv = np.ascontiguousarray(prev_mv, np.int16)
            prev_ref = np.ascontiguousarray(prev_ref, np.int8)
        lib().orc_set_ref(self.ctx, _p(y), _p(u), _p(v), _p(prev_mv), _p(prev_ref))

    def ref_planes(self):
        st, ln = lib().orc_ref_stride(self.ctx), lib().orc_ref_lines(self.ctx)
        out = np.zeros((4, ln, st), np.uint8)
        lib().orc_get_ref_planes(self.ctx, _p(out))
        return out

    def ref_integral(self):
        st, ln = lib().orc_ref_stride(self.ctx), lib().orc_ref_lines(self.ctx)
        out = np.zeros((ln, st), np.uint16)
        lib().orc_get_ref_integral(self.ctx, _p(out))
        return out

    def debug_state_hash(self, dump_mb=-1):
        self._dbg = np.zeros(self.n_mb, np.uint32)
        self._dump = np.zeros(460, np.uint8)
        lib().orc_set_debug(self.ctx, _p(self._dbg), dump_mb, _p(self._dump))
        return self._dbg

    def analyse_pframe(self, qp, embed=1):
        mbs = np.zeros(self.n_mb, MB_DTYPE)
        ry = np.zeros((self.h, self.w), np.uint8)
        ru = np.zeros((self.h // 2, self.w // 2), np.uint8)
        rv = np.zeros((self.h // 2, self.w // 2), np.uint8)
        rc = lib().orc_analyse_pframe(self.ctx, qp, embed, _p(mbs), _p(ry), _p(ru), _p(rv))
        if rc:
            raise RuntimeError("orc_analyse_pframe: %d (subme >= 6 with sub-8x8 partitions is not restated)" % rc)
        return mbs, (ry, ru, rv)

    def embed_pframe(self, mbs, emrate, message=None):
        cap = 16 * self.n_mb
        arr = dict(cover=np.zeros(cap, np.uint8), rho=np.zeros(cap, np.float32), message=np.zeros(cap, np.uint8),
                   stego=np.zeros(cap, np.uint8), flip=np.zeros(cap, np.int8))
        e = Embed(0, 0, 0, 0, *[arr[k].ctypes.data for k in ("cover", "rho", "message", "stego", "flip")])
        if message is not None:
            message = np.ascontiguousarray(message, np.uint8)
        lib().orc_embed_pframe(self.ctx, _p(mbs), C.c_float(emrate), _p(message),
                               0 if message is None else len(message), C.byref(e))
        out = {k: v[:e.n] for k, v in arr.items()}
        out["message"] = arr["message"][:e.m]
        out.update(n=e.n, m=e.m, stc_ok=e.stc_ok, num_flip=e.num_flip)
        return out

    def final_mvs(self, mbs, emb):
        cap = len(emb["flip"])
        flip = np.ascontiguousarray(emb["flip"], np.int8)
        e = Embed(emb["n"], emb["m"], emb["stc_ok"], emb["num_flip"], None, None, None, None, flip.ctypes.data)
        out = mbs.copy()
        lib().orc_final_mvs(self.ctx, C.byref(e), _p(out))
        return out

    def pass2_pframe(self, qp, mbs, flips):
        """final record, per-4x4 non-zero flags [n_mb, 16], reconstruction before / after the loop filter"""
        n = len(mbs)
        W, H = self.p.i_width, self.p.i_height
        flips = np.ascontiguousarray(flips, np.uint8)
        mbs = np.ascontiguousarray(mbs)
        out = np.zeros(n, MB_DTYPE)
        nnz = np.zeros((n, 16), np.uint8)
        planes = [np.zeros((H >> s, W >> s), np.uint8) for s in (0, 1, 1, 0, 1, 1)]
        lib().orc_pass2_pframe.restype = C.c_int
        k = lib().orc_pass2_pframe(self.ctx, qp, _p(mbs), _p(flips), len(flips), _p(out), _p(nnz), *[_p(a) for a in planes])
        if k < 0:
            raise RuntimeError("orc_pass2_pframe: flip map shorter than the carriers of the record")
        return out, nnz, tuple(planes[:3]), tuple(planes[3:]), k

    def me_search(self, qp, mb_x, mb_y, pixel, xoff, yoff, mvp, mvc):
        mvp = np.asarray(mvp, np.int16)
        mvc = np.ascontiguousarray(np.asarray(mvc, np.int16).reshape(-1, 2))
        mv = np.zeros(2, np.int16)
        cost = np.zeros(2, np.int32)
        lib().orc_me_search(self.ctx, qp, mb_x, mb_y, PIXEL[pixel], xoff, yoff, _p(mvp), _p(mvc), len(mvc), _p(mv), _p(cost))
        return mv, cost


def cost_mv_table(qp):
    out = np.zeros(4 * 4 * 2048 + 1, np.int16)
    lib().orc_cost_mv_table(qp, _p(out))
    return out


def stc_embed(cover, msg, rho, height=10):
    cover = np.ascontiguousarray(cover, np.uint8); msg = np.ascontiguousarray(msg, np.uint8)
    rho = np.ascontiguousarray(rho, np.float32)
    stego = np.zeros(len(cover), np.uint8)
    ok = lib().orc_stc_embed(_p(cover), len(cover), _p(msg), len(msg), _p(rho), _p(stego), height)
    return ok, stego


def stc_extract(stego, m, height=10):
    stego = np.ascontiguousarray(stego, np.uint8)
    msg = np.zeros(m, np.uint8)
    ok = lib().orc_stc_extract(_p(stego), len(stego), m, height, _p(msg))
    return ok, msg


def glibc_rand(n, seed=1):
    class R(C.Structure):
        _fields_ = [("r", C.c_int32 * 34), ("f", C.c_int), ("b", C.c_int)]
    s = R()
    lib().orc_srand(C.byref(s), seed)
    return np.array([lib().orc_rand(C.byref(s)) for _ in range(n)], np.int64)
