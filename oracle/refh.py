"""ctypes binding of oracle/_ref/libpcamv_ref.so (TEST INFRASTRUCTURE ONLY).

The library is the reference's own C sources compiled by oracle/Makefile plus
oracle/ref_harness.c.  Used by oracle/gen_golden.py to mint tests/golden/ fixtures and by
tests that cross-check the restatement wherever the prebuilt library is present.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libpcamv_ref.so")

ME = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}
PIXEL = {"16x16": 0, "16x8": 1, "8x16": 2, "8x8": 3, "8x4": 4, "4x8": 5, "4x4": 6}
P_L0, P_8x8, P_SKIP = 4, 5, 6
D_L0_4x4, D_L0_8x4, D_L0_4x8, D_L0_8x8 = 0, 1, 2, 3
D_8x8, D_16x8, D_8x16, D_16x16 = 13, 14, 15, 16


class RefMB(C.Structure):
    _fields_ = [("type", C.c_int32), ("partition", C.c_int32), ("qp", C.c_int32),
                ("sub_partition", C.c_uint8 * 4), ("ref", C.c_int8 * 16),
                ("mv", (C.c_int16 * 2) * 16), ("mv_stego", (C.c_int16 * 2) * 16),
                ("stego_cost", C.c_int32 * 16), ("pskip_mv", C.c_int16 * 2),
                ("mvr16", C.c_int16 * 2), ("used", C.c_uint8), ("pad", C.c_uint8 * 3)]


MB_DTYPE = np.dtype([("type", "<i4"), ("partition", "<i4"), ("qp", "<i4"),
                     ("sub_partition", "u1", (4,)), ("ref", "i1", (16,)),
                     ("mv", "<i2", (16, 2)), ("mv_stego", "<i2", (16, 2)),
                     ("stego_cost", "<i4", (16,)), ("pskip_mv", "<i2", (2,)),
                     ("mvr16", "<i2", (2,)), ("used", "u1"), ("pad", "u1", (3,))])
assert MB_DTYPE.itemsize == C.sizeof(RefMB)


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.refh_open.restype = C.c_void_p
        _lib.refh_open.argtypes = [C.c_int] * 12
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Ref:
    def __init__(self, width, height, qp=26, me="hex", me_range=16, subme=5, mv_range=128,
                 cabac=1, embed=1, inter_flags=-1, psy_rd=1.0, chroma_qp_offset=None):
        """psy_rd: --psy-rd strength (only acts from subme 6 on, encoder.c:513); chroma_qp_offset: the value the reference
        holds after x264_validate_parameters (default: 0, lowered by 2 when psy-RD is active, encoder.c:520-521)"""
        psy_fix8 = int(psy_rd * 256 + 0.5) if subme >= 6 else 0
        if chroma_qp_offset is None:
            chroma_qp_offset = 0 if not psy_fix8 else (-1 if psy_rd < 0.25 else -2)
        self.psy_fix8, self.chroma_qp_offset = psy_fix8, chroma_qp_offset
        self.w, self.h = width, height
        self.mb_w, self.mb_h = width // 16, height // 16
        self.qp = qp
        self.ctx = C.c_void_p(lib().refh_open(width, height, qp, ME[me], me_range, subme,
                                               mv_range, cabac, embed, inter_flags, psy_fix8, chroma_qp_offset))
        self._dbg = None
        if not self.ctx:
            raise RuntimeError("refh_open failed (size > 396 MBs?)")

    def debug_state_hash(self, dump_mb=-1):
        """ask for the FNV-1a hash of the CABAC context states after every macroblock of the following analyse calls"""
        self._dbg = np.zeros(self.mb_w * self.mb_h, np.uint32)
        self._dump = np.zeros(460, np.uint8)
        lib().refh_set_debug(self.ctx, _p(self._dbg), dump_mb, _p(self._dump))
        return self._dbg

    def set_ref(self, y, u, v, prev_mv=None, prev_ref=None, prev_type=None):
        y, u, v = [np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v)]
        if prev_mv is not None:
            prev_mv = np.ascontiguousarray(prev_mv, dtype=np.int16)
            prev_ref = np.ascontiguousarray(prev_ref, dtype=np.int8)
        if prev_type is not None:
            prev_type = np.ascontiguousarray(prev_type, dtype=np.int8)
        lib().refh_set_ref(self.ctx, _p(y), _p(u), _p(v), _p(prev_mv), _p(prev_ref), _p(prev_type))

    def set_fenc(self, y, u, v):
        y, u, v = [np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v)]
        lib().refh_set_fenc(self.ctx, _p(y), _p(u), _p(v))

    def ref_planes(self, want_integral=False):
        stride = lib().refh_ref_stride(self.ctx)
        lines = self.h + 64
        out = np.zeros((4, lines, stride), dtype=np.uint8)
        integ = np.zeros((lines, stride), dtype=np.uint16) if want_integral else None
        lib().refh_get_ref_planes(self.ctx, _p(out), _p(integ))
        return out, integ

    def analyse_pframe(self, qp=None):
        n = self.mb_w * self.mb_h
        mbs = np.zeros(n, dtype=MB_DTYPE)
        ry = np.zeros((self.h, self.w), np.uint8)
        ru = np.zeros((self.h // 2, self.w // 2), np.uint8)
        rv = np.zeros((self.h // 2, self.w // 2), np.uint8)
        lib().refh_analyse_pframe(self.ctx, self.qp if qp is None else qp, _p(mbs), _p(ry), _p(ru), _p(rv))
        return mbs, (ry, ru, rv)

    def pass2_pframe(self, flips, qp=None):
        """second pass + loop filter of the frame last analysed: final record, per-4x4 non-zero counts,
        reconstruction before / after deblocking; returns also the number of carrier MVs pass 2 walked"""
        n = self.mb_w * self.mb_h
        flips = np.ascontiguousarray(flips, dtype=np.int8)
        mbs = np.zeros(n, dtype=MB_DTYPE)
        nnz = np.zeros((n, 16), np.uint8)
        planes = [np.zeros((self.h >> s, self.w >> s), np.uint8) for s in (0, 1, 1, 0, 1, 1)]
        lib().refh_pass2_pframe.restype = C.c_int
        nmv = lib().refh_pass2_pframe(self.ctx, self.qp if qp is None else qp, _p(flips), len(flips), _p(mbs), _p(nnz), *[_p(a) for a in planes])
        if nmv < 0:
            raise RuntimeError("refh_pass2_pframe failed")
        return mbs, nnz, tuple(planes[:3]), tuple(planes[3:]), nmv

    def slice_data(self):
        """bytes the reference's entropy coder wrote for the frame analysed / re-encoded last: CABAC from the first mb_skip_flag on,
        CAVLC from the first mb_skip_run to the rbsp trailing bits"""
        buf = np.zeros(1 << 20, np.uint8)
        lib().refh_slice_data.restype = C.c_int
        n = lib().refh_slice_data(self.ctx, _p(buf), len(buf))
        if n < 0:
            raise RuntimeError(f"refh_slice_data: {n}")
        return buf[:n].tobytes()

    def me_search(self, qp, mb_x, mb_y, pixel, xoff, yoff, mvp, mvc):
        mvp = np.asarray(mvp, np.int16)
        mvc = np.ascontiguousarray(np.asarray(mvc, np.int16).reshape(-1, 2))
        mv = np.zeros(2, np.int16)
        cost = np.zeros(2, np.int32)
        lib().refh_me_search(self.ctx, qp, mb_x, mb_y, PIXEL[pixel], xoff, yoff, _p(mvp), _p(mvc),
                             len(mvc), _p(mv), _p(cost))
        return mv, cost


def cost_mv_table(qp):
    out = np.zeros(4 * 4 * 2048 + 1, np.int16)
    lib().refh_cost_mv_table.restype = C.c_int
    rc = lib().refh_cost_mv_table(qp, _p(out))
    return out, rc


def stc_embed(cover, msg, rho, height=10):
    cover = np.ascontiguousarray(cover, np.uint8)
    msg = np.ascontiguousarray(msg, np.uint8)
    rho = np.ascontiguousarray(rho, np.float32)
    stego = np.zeros(len(cover), np.uint8)
    lib().refh_stc_embed.restype = C.c_int
    ok = lib().refh_stc_embed(_p(cover), len(cover), _p(msg), len(msg), _p(rho), height, _p(stego))
    return ok, stego
