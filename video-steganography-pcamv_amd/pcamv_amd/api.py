import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(_PKG, "csrc")

ME_NAMES = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}      # x264_motion_est_names, x264.h:113
P_L0, P_8x8, P_SKIP = 4, 5, 6
I4x4, PSUB16x16, PSUB8x8 = 0x01, 0x10, 0x20


class PcamvError(RuntimeError):
    pass


class Params(C.Structure):
    """pcamv_params_t: the x264_param_t fields this path reads (same names)."""
    _fields_ = [("i_width", C.c_int32), ("i_height", C.c_int32), ("i_me_method", C.c_int32),
                ("i_me_range", C.c_int32), ("i_subpel_refine", C.c_int32), ("i_mv_range", C.c_int32),
                ("b_chroma_me", C.c_int32), ("b_fast_pskip", C.c_int32), ("b_dct_decimate", C.c_int32),
                ("b_cabac", C.c_int32), ("inter", C.c_uint32), ("i_chroma_qp_offset", C.c_int32),
                ("i_luma_deadzone", C.c_int32 * 2), ("i_tscale", C.c_int32), ("i_psy_rd", C.c_int32)]


class _Embed(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("stc_ok", C.c_int32), ("num_flip", C.c_int32),
                ("cover", C.c_void_p), ("rho", C.c_void_p), ("message", C.c_void_p),
                ("stego", C.c_void_p), ("flip", C.c_void_p)]


MB_DTYPE = np.dtype([("i_type", "<i4"), ("i_partition", "<i4"), ("i_qp", "<i4"),
                     ("i_sub_partition", "u1", (4,)), ("ref", "i1", (16,)),
                     ("mv", "<i2", (16, 2)), ("mv_stego", "<i2", (16, 2)),
                     ("inter_stego_cost", "<i4", (16,)), ("pskip_mv", "<i2", (2,)),
                     ("mvr16", "<i2", (2,)), ("used", "u1"), ("pad", "u1", (3,))])

_LEVELS = [(10, 1485, 99, 148500, 64), (11, 3000, 396, 337500, 128), (12, 6000, 396, 891000, 128),
           (13, 11880, 396, 891000, 128), (20, 11880, 396, 891000, 128), (21, 19800, 792, 1782000, 256),
           (22, 20250, 1620, 3037500, 256), (30, 40500, 1620, 3037500, 256), (31, 108000, 3600, 6750000, 512),
           (32, 216000, 5120, 7680000, 512), (40, 245760, 8192, 12288000, 512), (41, 245760, 8192, 12288000, 512),
           (42, 522240, 8704, 13056000, 512), (50, 589824, 22080, 41400000, 512), (51, 983040, 36864, 69120000, 512)]


def level_mv_range(width, height, fps=25):
    """Vertical MV range of the lowest H.264 level admitting the stream with one reference frame
    (what x264_validate_levels + encoder.c:540-559 leave in analyse.i_mv_range)."""
    mbs = (width // 16) * (height // 16)
    for _, mbps, fs, dpb, mvr in _LEVELS:
        if fs >= mbs and mbps >= mbs * fps and dpb >= 384 * mbs:
            return mvr
    return 512


def _validate(p):
    """the part of x264_validate_parameters that couples these fields (encoder.c:511-522): psy-RD acts from subme 6 on and
    lowers the chroma QP offset by 2 (by 1 below strength 0.25); the user's own values are kept on the side"""
    # (a Params() built directly, or a ctypes copy of one, has no side values: x264's defaults psy-rd 1.0 and the offset as it stands)
    if not hasattr(p, "_f_psy_rd"):
        p._f_psy_rd, p._chroma_qp_offset = 1.0, p.i_chroma_qp_offset
    f = p._f_psy_rd if p.i_subpel_refine >= 6 else 0.0
    p.i_psy_rd = int(min(max(f, 0.0), 10.0) * 256 + 0.5)
    off = p._chroma_qp_offset - ((1 if f < 0.25 else 2) if p.i_psy_rd else 0)
    p.i_chroma_qp_offset = min(max(off, -12), 12)
    return p


def param_default(width, height):
    """x264_param_default (common/common.c:39-146) for the fields of this path, then what x264_validate_parameters makes
    of them: subme 6 (RD mode decision), me hex, partitions p8x8 + i4x4, CABAC, psy-rd 1.0."""
    p = Params()
    p.i_width, p.i_height = width, height
    p.i_me_method, p.i_me_range, p.i_subpel_refine = ME_NAMES["hex"], 16, 6
    p.i_mv_range = level_mv_range(width, height)
    p.b_chroma_me = p.b_fast_pskip = p.b_dct_decimate = p.b_cabac = 1
    p.inter = I4x4 | PSUB16x16
    p.i_luma_deadzone[0], p.i_luma_deadzone[1] = 21, 11
    p.i_tscale = 256
    p._f_psy_rd, p._chroma_qp_offset = 1.0, 0
    return _validate(p)


def param_parse(p, name, value):
    """x264_param_parse (common/common.c:229-560) for the option names of this path."""
    name = name.lstrip("-").replace("_", "-")
    if name == "me":
        if value not in ME_NAMES:
            raise PcamvError(f"invalid value for me: {value}")
        p.i_me_method = ME_NAMES[value]
    elif name in ("merange", "me-range"):
        p.i_me_range = int(value)
    elif name in ("subme", "subq"):
        p.i_subpel_refine = int(value)
    elif name == "mvrange":
        p.i_mv_range = int(value)
    elif name in ("partitions", "analyse"):
        v = 0
        toks = [t.strip() for t in str(value).split(",")]
        if "none" in toks:
            v = 0
        if "all" in toks or "i4x4" in toks:
            v |= I4x4
        if "all" in toks or "p8x8" in toks:
            v |= PSUB16x16
        if "all" in toks or "p4x4" in toks:
            v |= PSUB8x8
        if not (v & PSUB16x16):
            v &= ~PSUB8x8
        p.inter = v
    elif name == "no-chroma-me":
        p.b_chroma_me = 0
    elif name == "no-fast-pskip":
        p.b_fast_pskip = 0
    elif name == "no-dct-decimate":
        p.b_dct_decimate = 0
    elif name == "no-cabac":
        p.b_cabac = 0
    elif name == "chroma-qp-offset":
        p._chroma_qp_offset = int(value)
    elif name == "psy-rd":
        p._f_psy_rd = float(str(value).split(":")[0])
    else:
        raise PcamvError(f"unknown option: {name}")
    return _validate(p)


def lib_path():
    # PCAMV_GPU_LIB: development override to A/B a differently compiled build of the same sources
    return os.environ.get("PCAMV_GPU_LIB") or os.path.join(_PKG, "libpcamv_gpu.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 of csrc/pcamv_gpu.hip (+ csrc/pcamv_tesa.hip, the --me tesa instance, and csrc/pcamv_rd.hip + csrc/pcamv_rd_lo.hip
    + csrc/pcamv_rd_spec{,2,4}.hip + csrc/pcamv_rd_tesa.hip, the six builds of the --subme 6 / 7 instance, compiled side by side) into the in-tree libpcamv_gpu.so."""
    out = lib_path()
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if not f.endswith(".o")]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "pcamv_gpu.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]
    objs, procs = [], []
    for unit in ("pcamv_gpu", "pcamv_tesa", "pcamv_rd", "pcamv_rd_lo", "pcamv_rd_spec", "pcamv_rd_spec2", "pcamv_rd_spec4", "pcamv_rd_tesa"):
        obj = os.path.join(_CSRC, unit + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["hipcc", *flags, "-c", "-o", obj, os.path.join(_CSRC, unit + ".hip")]))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), "hipcc -c (csrc/*.hip)")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, *objs])
    for obj in objs:
        os.remove(obj)
    return out


_lib = None


def load_library():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise PcamvError(f"{path} is missing: run __graft_entry__.build() (hipcc). There is no CPU fallback.")
        _lib = C.CDLL(path)
        _lib.pcamv_gpu_last_error.restype = C.c_char_p
        _lib.pcamv_gpu_last_error.argtypes = [C.c_void_p]
        _lib.pcamv_gpu_close.restype = None
        _lib.pcamv_gpu_close.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def parse_pslice_cabac(slice_data, mb_w, mb_h, qp):
    """pcamv_gpu_parse_pslice_cabac: the macroblock types, partitions and motion vectors a decoder reads out of a CABAC P slice"""
    data = np.frombuffer(bytes(slice_data), np.uint8)
    mbs = np.zeros(mb_w * mb_h, MB_DTYPE)
    lib = load_library()
    lib.pcamv_gpu_parse_pslice_cabac.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = lib.pcamv_gpu_parse_pslice_cabac(_p(data), len(data), mb_w, mb_h, qp, _p(mbs))
    if rc:
        raise PcamvError(f"pcamv_gpu_parse_pslice_cabac failed: {rc}")
    return mbs


def parse_pslice_cavlc(slice_data, mb_w, mb_h):
    """pcamv_gpu_parse_pslice_cavlc: the same out of a CAVLC P slice"""
    data = np.frombuffer(bytes(slice_data), np.uint8)
    mbs = np.zeros(mb_w * mb_h, MB_DTYPE)
    lib = load_library()
    lib.pcamv_gpu_parse_pslice_cavlc.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    rc = lib.pcamv_gpu_parse_pslice_cavlc(_p(data), len(data), mb_w, mb_h, _p(mbs))
    if rc:
        raise PcamvError(f"pcamv_gpu_parse_pslice_cavlc failed: {rc}")
    return mbs


def nal_to_rbsp(nal):
    """pcamv_gpu_nal_to_rbsp: (rbsp bytes, nal_ref_idc, nal_unit_type) of one NAL unit (Annex-B start code optional)"""
    data = np.frombuffer(bytes(nal), np.uint8)
    out = np.zeros(max(1, len(data)), np.uint8)
    n, ref_idc, typ = C.c_size_t(), C.c_int(), C.c_int()
    lib = load_library()
    lib.pcamv_gpu_nal_to_rbsp.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rc = lib.pcamv_gpu_nal_to_rbsp(_p(data), len(data), _p(out), C.byref(n), C.byref(ref_idc), C.byref(typ))
    if rc:
        raise PcamvError(f"pcamv_gpu_nal_to_rbsp failed: {rc}")
    return out[:n.value].tobytes(), ref_idc.value, typ.value


def parse_pslice_at(rbsp, start_bit, mb_w, mb_h, qp=None):
    """pcamv_gpu_parse_pslice_cabac_at (qp given) / _cavlc_at: the slice data behind a slice header that ends at bit start_bit of the RBSP"""
    data = np.frombuffer(bytes(rbsp), np.uint8)
    mbs = np.zeros(mb_w * mb_h, MB_DTYPE)
    lib = load_library()
    if qp is None:
        lib.pcamv_gpu_parse_pslice_cavlc_at.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        rc = lib.pcamv_gpu_parse_pslice_cavlc_at(_p(data), len(data), start_bit, mb_w, mb_h, _p(mbs))
    else:
        lib.pcamv_gpu_parse_pslice_cabac_at.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
        rc = lib.pcamv_gpu_parse_pslice_cabac_at(_p(data), len(data), start_bit, mb_w, mb_h, qp, _p(mbs))
    if rc:
        raise PcamvError(f"pcamv_gpu_parse_pslice_*_at failed: {rc}")
    return mbs


class StcLcg:
    """state of the reference's STC column generator (embed.h:134-139), carried from frame to frame by an extractor; a process --
    a closed GOP under the per-GOP parity definition -- starts at 1"""

    def __init__(self, state=1):
        self.state = C.c_int64(state)


def stc_extract(stego, m, height=10, lcg=None):
    """message bits out of the stego bits; lcg (StcLcg) is needed, and advanced, for sub-matrix widths outside 2..20"""
    stego = np.ascontiguousarray(stego, np.uint8)
    msg = np.zeros(m, np.uint8)
    rc = load_library().pcamv_gpu_stc_extract_lcg(_p(stego), len(stego), m, height, C.byref(lcg.state) if lcg is not None else None, _p(msg))
    if rc:
        raise PcamvError(f"pcamv_gpu_stc_extract failed: {rc}")
    return msg


class Encoder:
    """One analysis context = one x264_t's worth of P-frame analysis state on one GPU."""

    def __init__(self, params, device=0):
        self.lib = load_library()
        self.p = params
        self.w, self.h = params.i_width, params.i_height
        self.n_mb = (self.w // 16) * (self.h // 16)
        ctx = C.c_void_p()
        rc = self.lib.pcamv_gpu_open(C.byref(params), device, C.byref(ctx))
        if rc:
            names = {-1: "invalid parameter", -2: "no HIP device (there is no CPU fallback)", -3: "out of memory",
                     -4: "HIP error", -5: "unsupported (subme >= 8, tesa with me_range > 16, ... are not on the GPU path)"}
            raise PcamvError(f"pcamv_gpu_open failed: {names.get(rc, rc)}")
        self.ctx = ctx

    def _chk(self, rc, what):
        if rc:
            raise PcamvError(f"{what} failed ({rc}): {self.lib.pcamv_gpu_last_error(self.ctx).decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.pcamv_gpu_close(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _planes(self, y, u, v):
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (y, u, v)]
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs])
        strides = (C.c_int * 3)(*[a.shape[1] for a in arrs])
        return arrs, ptrs, strides

    def upload_fenc(self, y, u, v):
        keep, ptrs, strides = self._planes(y, u, v)
        self._chk(self.lib.pcamv_gpu_upload_fenc(self.ctx, ptrs, strides), "upload_fenc")

    def set_ref(self, y, u, v, prev_mv=None, prev_ref=None):
        keep, ptrs, strides = self._planes(y, u, v)
        if prev_mv is not None:
            prev_mv = np.ascontiguousarray(prev_mv, np.int16)
            prev_ref = np.ascontiguousarray(prev_ref, np.int8)
        self._chk(self.lib.pcamv_gpu_set_ref(self.ctx, ptrs, strides, _p(prev_mv), _p(prev_ref)), "set_ref")

    def ref_planes(self):
        stride = (self.w + 64 + 15) & ~15
        out = np.zeros((4, self.h + 64, stride), np.uint8)
        st, ln = C.c_int(), C.c_int()
        self._chk(self.lib.pcamv_gpu_get_ref_planes(self.ctx, _p(out), C.byref(st), C.byref(ln)), "get_ref_planes")
        assert st.value == stride and ln.value == self.h + 64
        return out

    def analyse_pframe(self, qp, embed=1, want_recon=True):
        mbs = np.zeros(self.n_mb, MB_DTYPE)
        rec = [np.zeros((self.h, self.w), np.uint8), np.zeros((self.h // 2, self.w // 2), np.uint8),
               np.zeros((self.h // 2, self.w // 2), np.uint8)]
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in rec]) if want_recon else None
        self._chk(self.lib.pcamv_gpu_analyse_pframe(self.ctx, qp, embed, _p(mbs), ptrs), "analyse_pframe")
        return mbs, rec

    def _embed_bufs(self):
        cap = 16 * self.n_mb
        arr = dict(cover=np.zeros(cap, np.uint8), rho=np.zeros(cap, np.float32), message=np.zeros(cap, np.uint8),
                   stego=np.zeros(cap, np.uint8), flip=np.zeros(cap, np.int8))
        e = _Embed(0, 0, 0, 0, *[arr[k].ctypes.data for k in ("cover", "rho", "message", "stego", "flip")])
        return arr, e

    @staticmethod
    def _embed_out(arr, e):
        out = {k: v[:e.n].copy() for k, v in arr.items()}
        out["message"] = arr["message"][:e.m].copy()
        out.update(n=e.n, m=e.m, stc_ok=e.stc_ok, num_flip=e.num_flip)
        return out

    def embed_pframe(self, emrate, message=None):
        arr, e = self._embed_bufs()
        if message is not None:
            message = np.ascontiguousarray(message, np.uint8)
        self._chk(self.lib.pcamv_gpu_embed_pframe(self.ctx, C.c_float(emrate), _p(message),
                                                  0 if message is None else len(message), C.byref(e)), "embed_pframe")
        return self._embed_out(arr, e)

    def final_mvs(self, mbs):
        out = mbs.copy()
        self._chk(self.lib.pcamv_gpu_final_mvs(self.ctx, _p(out)), "final_mvs")
        return out

    def pass2_pframe(self, flips=None):
        """pass 2 + loop filter of the frame last analysed: (final record, reconstruction, deblocked picture);
        flips = None uses the flip map of the last embed_pframe"""
        out = np.zeros(self.n_mb, MB_DTYPE)
        W, H = self.p.i_width, self.p.i_height
        planes = [np.zeros((H >> s, W >> s), np.uint8) for s in (0, 1, 1, 0, 1, 1)]
        rec = (C.c_void_p * 3)(*[a.ctypes.data for a in planes[:3]])
        dbk = (C.c_void_p * 3)(*[a.ctypes.data for a in planes[3:]])
        if flips is not None:
            flips = np.ascontiguousarray(flips, np.uint8)
        self._chk(self.lib.pcamv_gpu_pass2_pframe(self.ctx, _p(flips), 0 if flips is None else len(flips), _p(out), rec, dbk), "pass2_pframe")
        return out, tuple(planes[:3]), tuple(planes[3:])

    def debug_state_hash(self, enable=True):
        """diagnostics: FNV-1a of the CABAC context states after every macroblock of the following analyses (--subme >= 6)"""
        self._chk(self.lib.pcamv_gpu_debug_state_hash(self.ctx, int(enable)), "debug_state_hash")

    def state_hash_fetch(self):
        out = np.zeros(self.n_mb, np.uint32)
        self._chk(self.lib.pcamv_gpu_debug_state_hash_fetch(self.ctx, _p(out)), "debug_state_hash_fetch")
        return out

    def block_costs(self, qp, requests):
        req = np.ascontiguousarray(requests, np.int32).reshape(-1, 8)
        out = np.zeros((len(req), 3), np.int32)
        self._chk(self.lib.pcamv_gpu_block_costs(self.ctx, qp, len(req), _p(req), _p(out)), "block_costs")
        return out

    def rd_probe(self, qp, requests):
        """pcamv_gpu_rd_probe: requests = uint8 [n, 1024] (layout in include/pcamv_gpu.h) -> int32 [n, 32]"""
        req = np.ascontiguousarray(requests, np.uint8).reshape(-1, 1024)
        out = np.zeros((len(req), 32), np.int32)
        self.lib.pcamv_gpu_rd_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._chk(self.lib.pcamv_gpu_rd_probe(self.ctx, qp, len(req), _p(req), _p(out)), "rd_probe")
        return out

    def trace_mb(self, mb):
        self._chk(self.lib.pcamv_gpu_trace_mb(self.ctx, mb), "trace_mb")

    def trace_fetch(self):
        out = np.zeros(1 + 8 * 4000, np.int32)
        self._chk(self.lib.pcamv_gpu_trace_fetch(self.ctx, _p(out)), "trace_fetch")
        return out[1:1 + 8 * out[0]].reshape(-1, 8)

    # device-resident path (bench.py): raw device pointers as integers
    PREV_INTERNAL = 1

    def set_ref_device(self, y, u, v, prev_mv=0, prev_ref=0):
        self._chk(self.lib.pcamv_gpu_set_ref_device(self.ctx, C.c_void_p(y), C.c_void_p(u), C.c_void_p(v),
                                                    C.c_void_p(prev_mv or None), C.c_void_p(prev_ref or None)), "set_ref_device")

    def recon_device(self):
        """device pointers of the context's reconstruction planes (after a closed-loop step: the deblocked picture)"""
        out = (C.c_void_p * 3)()
        self._chk(self.lib.pcamv_gpu_recon_device(self.ctx, out), "recon_device")
        return [int(out[i]) for i in range(3)]

    def fetch_recon(self):
        W, H = self.p.i_width, self.p.i_height
        planes = [np.zeros((H >> s, W >> s), np.uint8) for s in (0, 1, 1)]
        arr = (C.c_void_p * 3)(*[a.ctypes.data for a in planes])
        self._chk(self.lib.pcamv_gpu_fetch_recon(self.ctx, arr), "fetch_recon")
        return tuple(planes)

    def set_fenc_device(self, y, u, v):
        self._chk(self.lib.pcamv_gpu_set_fenc_device(self.ctx, C.c_void_p(y), C.c_void_p(u), C.c_void_p(v)), "set_fenc_device")

    def step_device(self, qp, emrate, stream=0):
        self._chk(self.lib.pcamv_gpu_step_device(self.ctx, qp, C.c_float(emrate), C.c_void_p(stream or None)), "step_device")

    def fetch_results(self, want_embed=True):
        mbs = np.zeros(self.n_mb, MB_DTYPE)
        arr, e = self._embed_bufs()
        self._chk(self.lib.pcamv_gpu_fetch_results(self.ctx, _p(mbs), C.byref(e) if want_embed else None), "fetch_results")
        return mbs, (self._embed_out(arr, e) if want_embed else None)

    def kernel_time(self, kernel=None, reset=True):
        """average launch time of the analysis kernel of the active schedule (k_analyse_flow, or k_search_diag under PCAMV_SCHED=diag)"""
        ms, n = C.c_double(), C.c_int()
        names = [kernel] if kernel else ["k_analyse_flow", "k_search_diag"]
        for nm in names:
            rc = self.lib.pcamv_gpu_kernel_time(self.ctx, nm.encode(), C.byref(ms), C.byref(n), int(reset))
            if rc == 0:
                break
        self._chk(rc, "kernel_time")
        return ms.value, n.value


class Batch:
    """Several Encoder contexts (independent closed GOPs) stepped together: pcamv_gpu_batch_*."""

    def __init__(self, encoders):
        self.lib = load_library()
        self.encs = list(encoders)
        arr = (C.c_void_p * len(self.encs))(*[e.ctx.value for e in self.encs])
        b = C.c_void_p()
        rc = self.lib.pcamv_gpu_batch_create(arr, len(self.encs), C.byref(b))
        if rc:
            raise PcamvError(f"pcamv_gpu_batch_create failed: {rc}")
        self.b = b
        self.lib.pcamv_gpu_batch_last_error.restype = C.c_char_p
        self.lib.pcamv_gpu_batch_last_error.argtypes = [C.c_void_p]
        self.lib.pcamv_gpu_batch_destroy.restype = None
        self.lib.pcamv_gpu_batch_destroy.argtypes = [C.c_void_p]

    def step(self, qp, emrate, stream=0):
        rc = self.lib.pcamv_gpu_batch_step(self.b, qp, C.c_float(emrate), C.c_void_p(stream or None))
        if rc:
            raise PcamvError(f"batch_step failed ({rc}): {self.lib.pcamv_gpu_batch_last_error(self.b).decode()}")

    def set_closed_loop(self, on=True):
        if self.lib.pcamv_gpu_batch_set_closed_loop(self.b, int(on)):
            raise PcamvError("batch_set_closed_loop failed")

    def copy_results_async(self, dst_mb, mb_stride, dst_flip=0, flip_stride=0, stream=0):
        """records (and flip maps) of the step enqueued last to device or pinned host memory, on `stream`, without a host sync"""
        rc = self.lib.pcamv_gpu_batch_copy_results_async(self.b, C.c_void_p(dst_mb), C.c_size_t(mb_stride), C.c_void_p(dst_flip or None),
                                                         C.c_size_t(flip_stride), C.c_void_p(stream or None))
        if rc:
            raise PcamvError(f"batch_copy_results_async failed ({rc}): {self.lib.pcamv_gpu_batch_last_error(self.b).decode()}")

    def dominant_kernel(self):
        self.lib.pcamv_gpu_batch_dominant_kernel.restype = C.c_char_p
        return self.lib.pcamv_gpu_batch_dominant_kernel(self.b).decode()

    def kernel_time(self, kernel=None, reset=True):
        ms, n = C.c_double(), C.c_int()
        kernel = kernel or self.dominant_kernel()
        rc = self.lib.pcamv_gpu_batch_kernel_time(self.b, kernel.encode(), C.byref(ms), C.byref(n), int(reset))
        if rc:
            raise PcamvError(f"batch_kernel_time failed: {rc}")
        return ms.value, n.value

    def close(self):
        if getattr(self, "b", None):
            self.lib.pcamv_gpu_batch_destroy(self.b)
            self.b = None
