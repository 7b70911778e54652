"""Host-side mirror (Python) of the C ABI in include/pcamv_gpu.h.

The compute is the HIP library libpcamv_gpu.so; this package only moves buffers and mirrors the
reference's parameter surface for this path (x264_param_default / x264_param_parse names).
There is no CPU fallback: importing works without a GPU, but every operation raises
PcamvError when the library or a HIP device is missing.
"""
from .api import (PcamvError, Params, Encoder, Batch, param_default, param_parse, level_mv_range, lib_path, build_library,
                  MB_DTYPE, stc_extract, StcLcg, parse_pslice_cabac, parse_pslice_cavlc, parse_pslice_at, nal_to_rbsp, load_library, ME_NAMES, P_L0, P_8x8, P_SKIP)

__all__ = ["PcamvError", "Params", "Encoder", "Batch", "param_default", "param_parse", "level_mv_range", "lib_path",
           "build_library", "MB_DTYPE", "stc_extract", "StcLcg", "parse_pslice_cabac", "parse_pslice_cavlc", "parse_pslice_at", "nal_to_rbsp", "load_library", "ME_NAMES", "P_L0", "P_8x8", "P_SKIP"]
