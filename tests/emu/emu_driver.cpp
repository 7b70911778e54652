/*
 * emu_driver.cpp -- TEST-ONLY CPU build of the product's wave-uniform control code
 * (csrc/pcamv_logic.h + pcamv_mbkernels.h) with scalar primitives, driven in raster order.
 * Lets `pytest -m "not gpu"` (and sanitizers) check the search / decision / RCA logic that the
 * HIP kernels execute, without a GPU.  It is NOT a fallback: libpcamv_gpu.so never links it.
 */
#define PCAMV_HOST_EMU 1
#include <stdlib.h>
#include <stdio.h>
#include "pcamv_common.h"
#include "pcamv_prims_emu.h"
#include "pcamv_mbkernels.h"
#include "pcamv_host_tables.h"

/* the instance the library would run: the --me tesa one (variant bit 0) only for that method; bit 1 the RD mode decision, bit 3 x264_rd_cost_part */
#define EMU_SEARCH(F, L, a, x, y) do { if ((F).me_method == PCAMV_ME_TESA) mbk_search<11>(F, L, a, x, y); else mbk_search<10>(F, L, a, x, y); } while (0)
extern "C" int emu_analyse_pframe(const pcamv_params_t *p, int qp, int embed,
                                  const uint8_t *fy, const uint8_t *fu, const uint8_t *fv,
                                  uint8_t *luma4, uint8_t *cu, uint8_t *cv,
                                  const int16_t *prev_mv, const int8_t *prev_ref,
                                  pcamv_mb_t *out, uint8_t *ry, uint8_t *ru, uint8_t *rv, int diag_order, int *trace, int trace_mb, uint32_t *dbg_hash)
{
    FrameDev F = {};
    pcamv_frame_set_params(&F, p);
    pcamv_frame_set_qp(&F, p, qp);
    F.embed = embed; F.trace = trace; F.trace_mb = trace_mb;
    F.fenc[0] = fy; F.fenc[1] = fu; F.fenc[2] = fv;
    size_t lsz = (size_t)F.stride * F.lines;
    for (int k = 0; k < 4; k++) F.luma[k] = luma4 + k * lsz + (size_t)F.stride * PCAMV_PAD + PCAMV_PAD;
    F.chroma[0] = cu + (size_t)F.cstride * PCAMV_CPAD + PCAMV_CPAD;
    F.chroma[1] = cv + (size_t)F.cstride * PCAMV_CPAD + PCAMV_CPAD;
    F.rec[0] = ry; F.rec[1] = ru; F.rec[2] = rv;
    F.mb_type = (int8_t *)malloc(F.n_mb);
    F.mv = (int16_t *)calloc((size_t)F.n_mb * 32, 2);
    F.ref8 = (int8_t *)malloc((size_t)F.n_mb * 4);
    F.mvr = (int16_t *)calloc((size_t)F.n_mb * 2, 2);
    F.mvp_aux = (int16_t *)calloc((size_t)F.n_mb * 32, 2);
    F.prev_mv = prev_mv; F.prev_ref = prev_ref; F.have_prev = prev_mv != NULL && p->i_tscale != 0;
    F.rec_mb = out;
    /* --subme >= 6 */
    F.ref_is_inter = prev_mv != NULL;
    F.nb_nz = (uint8_t *)calloc((size_t)F.n_mb, 16); F.nb_cbp = (int16_t *)calloc((size_t)F.n_mb, 2); F.nb_mvd = (int16_t *)calloc((size_t)F.n_mb * 16, 2);
    uint8_t cab[PCAMV_CHAIN_BYTES] = {0}, cab_init[PCAMV_CHAIN_BYTES] = {0}; uint32_t cab_tab[256];
    pcamv_build_cabac_init(qp, cab_init); pcamv_build_cabac_tab(cab_tab);
    F.cabac = cab; F.cabac_init = cab_init; F.cabac_tab = cab_tab; F.dbg_hash = dbg_hash;
    int16_t *cost = (int16_t *)malloc(PCAMV_COST_MV_LEN * sizeof(int16_t));
    pcamv_build_cost_mv(qp, cost);
    F.cost_mv = cost + PCAMV_COST_MV_CENTRE;
    MBLocal *L = (MBLocal *)malloc(sizeof(MBLocal));
    Analysis *a = (Analysis *)malloc(sizeof(Analysis));
    if (diag_order == 3) {  /* raster order, fused: what the dataflow schedule does when the entropy coder is CABAC (one chain per frame) */
        for (int y = 0; y < F.mb_h; y++) for (int x = 0; x < F.mb_w; x++) { EMU_SEARCH(F, L, a, x, y); mbk_rca_encode(F, L, a, y * F.mb_w + x, 1, F.b_mbrd); }
        free(L); free(a); free(cost); free(F.mb_type); free(F.mv); free(F.ref8); free(F.mvr); free(F.mvp_aux); free(F.nb_nz); free(F.nb_cbp); free(F.nb_mvd);
        return 0;
    }
    if (diag_order == 2) {  /* dataflow schedule: search, then RCA + reconstruction of the same macroblock, in a dependency-legal order */
        for (int d = 0; d < F.mb_w + 2 * (F.mb_h - 1); d++)
            for (int y = F.mb_h - 1; y >= 0; y--) { int x = d - 2 * y; if (x >= 0 && x < F.mb_w) { EMU_SEARCH(F, L, a, x, y); mbk_rca_encode(F, L, a, y * F.mb_w + x, 1, F.b_mbrd); } }
        free(L); free(a); free(cost); free(F.mb_type); free(F.mv); free(F.ref8); free(F.mvr); free(F.mvp_aux);
        return 0;
    }
    if (diag_order) {       /* the order the GPU uses: anti-diagonals x + 2y = d */
        for (int d = 0; d < F.mb_w + 2 * (F.mb_h - 1); d++)
            for (int y = 0; y < F.mb_h; y++) { int x = d - 2 * y; if (x >= 0 && x < F.mb_w) EMU_SEARCH(F, L, a, x, y); }
    } else
        for (int y = 0; y < F.mb_h; y++) for (int x = 0; x < F.mb_w; x++) EMU_SEARCH(F, L, a, x, y);
    if (embed)
        for (int xy = F.n_mb - 1; xy >= 0; xy--) for (int k = 15; k >= 0; k--) mbk_rca(F, L, a, xy, k);
    for (int xy = 0; xy < F.n_mb; xy++) mbk_encode(F, L, a, xy);
    free(L); free(a); free(cost); free(F.mb_type); free(F.mv); free(F.ref8); free(F.mvr); free(F.mvp_aux);
    return 0;
}

extern "C" void emu_get_stats(long long *out, int reset) { for (int i = 0; i < 32; i++) { out[i] = emu_stats[i]; if (reset) emu_stats[i] = 0; } }

/* The strip layout of the device's luma planes (pcamv_common.h): the host-side statement of its address arithmetic, for
 * tests/test_logic_emu.py::test_strip_layout_arithmetic.  Returns the first x for which the multiply-shift strip index differs
 * from x / 28 (or -1), and fills a small raster -> strip -> raster round trip. */
extern "C" int emu_strip_index_first_bad(int limit)
{
    for (int x = 0; x < limit; x++)
        if ((int)PCAMV_LSTRIP_OF(x) != x / PCAMV_LSW) return x;
    return -1;
}
extern "C" long long emu_strip_offset(int x, int y, int lines)
{
    const long long lskip = (long long)PCAMV_LROW * lines - PCAMV_LSW;
    return (long long)y * PCAMV_LROW + x + (long long)PCAMV_LSTRIP_OF(x) * lskip;
}
extern "C" long long emu_strip_plane_size(int stride, int lines) { return (long long)PCAMV_LSTRIPS(stride) * PCAMV_LROW * lines; }
