"""The reference-side binding (integration/pcamv_x264_glue.c, what INTEGRATION.md describes) must go through a compiler
against the reference's own headers and include/pcamv_gpu.h.  Build-container only: the GPU box has no /root/reference."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "encoder")), reason="needs the reference's headers (/root/reference)")
def test_glue_compiles_against_the_reference_headers():
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           f"-I{REF}", f"-I{REF}/common", f"-I{REF}/encoder", "-DHAVE_MALLOC_H", "-DARCH_X86_64", "-DSYS_LINUX",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "pcamv_x264_glue.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    errors = [l for l in r.stderr.splitlines() if "pcamv_x264_glue.c" in l and "error" in l]
    assert r.returncode == 0 and not errors, r.stderr[-3000:]


def test_integration_md_quotes_the_glue_verbatim():
    """INTEGRATION.md's blocks marked `<!-- verbatim: FILE -->` are text of FILE, character for character (runs anywhere)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    parts = md.split("<!-- verbatim: ")[1:]
    assert len(parts) >= 4
    for part in parts:
        path, rest = part.split(" -->", 1)
        body = rest.split("```c\n", 1)[1].split("\n```", 1)[0]
        src = open(os.path.join(ROOT, path)).read()
        assert body in src, f"INTEGRATION.md block not found in {path}: {body[:80]!r}"
    for stale in ("iGpu", "force_from_record", "h->gpu"):
        assert stale not in md
