"""-m 'not gpu': the C-ABI library loads and exports every symbol include/pcamv_gpu.h declares;
no compute is called (there is no GPU here) and opening a context must fail loudly, not fall back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    import pcamv_amd
    if not os.path.exists(pcamv_amd.lib_path()):
        pcamv_amd.build_library()
    lib = ctypes.CDLL(pcamv_amd.lib_path())
    hdr = open(os.path.join(ROOT, "include", "pcamv_gpu.h")).read()
    names = sorted(set(re.findall(r"\b(pcamv_gpu_\w+)\s*\(", hdr)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pcamv_gpu.h but not exported"
    assert lib.pcamv_gpu_abi_version() == 3     # round 3: rd_probe, NAL / bit-offset slice parsing, the speculative raster instance


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pcamv_amd
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.Encoder(pcamv_amd.param_default(176, 144))


def test_product_does_not_reference_the_oracle():
    """the product sources must never include, link or load anything under oracle/"""
    pkg = os.path.join(ROOT, "video-steganography-pcamv_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".h", ".hip", ".py", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "import orc" not in txt and "pcamv_oracle" not in txt, os.path.join(dp, f)


def test_param_parse_mirrors_reference_option_names():
    import pcamv_amd
    p = pcamv_amd.param_default(1920, 1088)
    # x264_param_default (common/common.c:83-140) as main() leaves it: hex, range 16, subme 6, i4x4 + p8x8, psy-rd 1.0 -> 256 and
    # the chroma QP offset it implies at subme >= 6 (encoder.c:513-521)
    assert (p.i_me_method, p.i_me_range, p.i_subpel_refine, p.inter, p.i_mv_range) == (1, 16, 6, 0x11, 512)
    assert (p.i_psy_rd, p.i_chroma_qp_offset, p.b_cabac) == (256, -2, 1)
    pcamv_amd.param_parse(p, "--me", "umh"); pcamv_amd.param_parse(p, "subme", 4); pcamv_amd.param_parse(p, "partitions", "p8x8,p4x4")
    assert (p.i_me_method, p.i_subpel_refine, p.inter) == (2, 4, 0x30)
    with pytest.raises(pcamv_amd.PcamvError):
        pcamv_amd.param_parse(p, "me", "bogus")
    assert pcamv_amd.level_mv_range(352, 288) == 128 and pcamv_amd.level_mv_range(176, 144) == 128
