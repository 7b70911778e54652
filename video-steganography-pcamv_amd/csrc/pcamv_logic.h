/*
 * pcamv_logic.h -- wave-uniform control code of the P-frame analysis: neighbour context and MV
 * prediction, integer/sub-pel motion search, partition decision, P_SKIP probe, macroblock
 * re-encode and the replacement-MV ("RCA") cost.  One wavefront runs this for one macroblock:
 * every value here is wave-uniform, the per-pixel work is in the prim_* functions
 * (pcamv_prims_gpu.h: lane-parallel; pcamv_prims_emu.h: scalar, used only by tests/ to run this
 * same control code on the CPU).
 *
 * Reference behaviour restated (file:line of /root/reference):
 *   mb_load            common/macroblock.c:914-1238, encoder/analyse.c:268-318
 *   predict_mv*        common/macroblock.c:28-163, 388-470
 *   me_search          encoder/me.c:158-666       refine_subpel  encoder/me.c:715-843
 *   analyse_*          encoder/analyse.c:1122-1693, 2613-2827, 3518-3689
 *   probe_pskip        encoder/macroblock.c:809-895
 *   mb_encode          encoder/macroblock.c:277-372, 484-802
 *   rca_mv_cost        encoder/analyse.c:2364-2550
 */
#ifndef PCAMV_LOGIC_H
#define PCAMV_LOGIC_H
#include "pcamv_common.h"

/* ---------------------------------------------------------------- neighbour context */
/* x264_macroblock_cache_mv / _ref (common/macroblock.h): fill a w x h patch of 4x4 positions; one
 * position per lane */
PCAMV_DEV void cache_mv_set(MBLocal *L, int x, int y, int w, int h, int mvx, int mvy)
{
    PCAMV_WAVE_SYNC();
    FOR_CAND(i, w * h) {
        int j = i / w, k = i - j * w, idx = SCAN8_0 + x + k + 8 * (y + j);
        L->cmv[idx][0] = (int16_t)mvx; L->cmv[idx][1] = (int16_t)mvy;
    }
    PCAMV_WAVE_SYNC();
}
PCAMV_DEV void cache_ref_set(MBLocal *L, int x, int y, int w, int h, int ref)
{
    PCAMV_WAVE_SYNC();
    FOR_CAND(i, w * h) { int j = i / w, k = i - j * w; L->cref[SCAN8_0 + x + k + 8 * (y + j)] = (int8_t)ref; }
    PCAMV_WAVE_SYNC();
}

PCAMV_DEV void predict_from3(int ref, int refa, int refb, int refc, const int16_t *a, const int16_t *b, const int16_t *c, int mvp[2])
{
    int cnt = (refa == ref) + (refb == ref) + (refc == ref);
    if (cnt > 1) { mvp[0] = median3i(a[0], b[0], c[0]); mvp[1] = median3i(a[1], b[1], c[1]); }
    else if (cnt == 1) {
        const int16_t *s = refa == ref ? a : refb == ref ? b : c;
        mvp[0] = s[0]; mvp[1] = s[1];
    } else if (refb == -2 && refc == -2 && refa != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = median3i(a[0], b[0], c[0]); mvp[1] = median3i(a[1], b[1], c[1]); }
}
PCAMV_DEV void predict_mv(MBLocal *L, int idx, int width, int mvp[2])
{
    int i8 = scan8_of(idx);
    int ref = L->cref[i8];
    int refa = L->cref[i8 - 1], refb = L->cref[i8 - 8], refc = L->cref[i8 - 8 + width];
    const int16_t *a = L->cmv[i8 - 1], *b = L->cmv[i8 - 8], *c = L->cmv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || refc == -2) { refc = L->cref[i8 - 8 - 1]; c = L->cmv[i8 - 8 - 1]; }
    if (L->i_partition == PCAMV_D_16x8) {
        if (idx == 0 && refb == ref) { mvp[0] = b[0]; mvp[1] = b[1]; return; }
        if (idx != 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
    } else if (L->i_partition == PCAMV_D_8x16) {
        if (idx == 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
        if (idx != 0 && refc == ref) { mvp[0] = c[0]; mvp[1] = c[1]; return; }
    }
    predict_from3(ref, refa, refb, refc, a, b, c, mvp);
}
PCAMV_DEV void predict_mv_16x16(MBLocal *L, int ref, int mvp[2])
{
    int refa = L->cref[SCAN8_0 - 1], refb = L->cref[SCAN8_0 - 8], refc = L->cref[SCAN8_0 - 8 + 4];
    const int16_t *a = L->cmv[SCAN8_0 - 1], *b = L->cmv[SCAN8_0 - 8], *c = L->cmv[SCAN8_0 - 8 + 4];
    if (refc == -2) { refc = L->cref[SCAN8_0 - 8 - 1]; c = L->cmv[SCAN8_0 - 8 - 1]; }
    predict_from3(ref, refa, refb, refc, a, b, c, mvp);
}
PCAMV_DEV void predict_mv_pskip(MBLocal *L, int mv[2])
{
    int refa = L->cref[SCAN8_0 - 1], refb = L->cref[SCAN8_0 - 8];
    const int16_t *a = L->cmv[SCAN8_0 - 1], *b = L->cmv[SCAN8_0 - 8];
    if (refa == -2 || refb == -2 || !(refa | a[0] | a[1]) || !(refb | b[0] | b[1])) { mv[0] = mv[1] = 0; }
    else predict_mv_16x16(L, 0, mv);
}
/* candidate MVs for the 16x16 search: spatial 16x16 results of non-skipped neighbours, then the
 * co-located / right / below MVs of the previous frame scaled by POC distance */
PCAMV_DEV int predict_mv_ref16x16(const FrameDev &F, MBLocal *L, int (*mvc)[2])
{
    int i = 0, xy = L->mb_xy, top = xy - F.mb_w;
    /* every candidate is requested before the first is used (a neighbour that does not count reads this macroblock's own slot):
     * one memory round trip for the four spatial ones and one for the temporal ones, instead of one per candidate */
    const int use_l = (L->neighbour & NB_LEFT) && L->type_left != PCAMV_P_SKIP;
    const int use_t = (L->neighbour & NB_TOP) && L->type_top != PCAMV_P_SKIP;
    const int use_tl = (L->neighbour & NB_TOP) && (L->neighbour & NB_TOPLEFT) && L->type_topleft != PCAMV_P_SKIP;
    const int use_tr = (L->neighbour & NB_TOP) && L->mb_x < F.mb_w - 1 && L->type_topright != PCAMV_P_SKIP;
    const uint32_t w_l = NB_LD32(&F.mvr[2 * (use_l ? xy - 1 : xy)]), w_t = NB_LD32(&F.mvr[2 * (use_t ? top : xy)]);
    const uint32_t w_tl = NB_LD32(&F.mvr[2 * (use_tl ? top - 1 : xy)]), w_tr = NB_LD32(&F.mvr[2 * (use_tr ? top + 1 : xy)]);
#define SETMV(w_) { mvc[i][0] = (int16_t)((w_) & 0xffff); mvc[i][1] = (int16_t)((w_) >> 16); i++; }
    if (use_l) SETMV(w_l);
    if (use_t) SETMV(w_t);
    if (use_tl) SETMV(w_tl);
    if (use_tr) SETMV(w_tr);
#undef SETMV
    if (F.have_prev) {
        int ok[3], rf[3]; uint32_t mw[3];
        for (int k = 0; k < 3; k++) {
            const int dx = k == 1, dy = k == 2;
            ok[k] = !(k == 1 && !(L->mb_x < F.mb_w - 1)) && !(k == 2 && !(L->mb_y < F.mb_h - 1));
            const int b4 = 4 * (L->mb_y * 4 * F.mb_w + L->mb_x) + (ok[k] ? dx * 4 + dy * 4 * (4 * F.mb_w) : 0);
            const int b8 = 2 * (L->mb_y * 2 * F.mb_w + L->mb_x) + (ok[k] ? dx * 2 + dy * 2 * (2 * F.mb_w) : 0);
            rf[k] = F.prev_ref[b8];
            mw[k] = *(const uint32_t *)&F.prev_mv[2 * b4];
        }
        for (int k = 0; k < 3; k++)
            if (ok[k] && rf[k] >= 0) {
                mvc[i][0] = (int16_t)(((int)(int16_t)(mw[k] & 0xffff) * F.tscale + 128) >> 8);
                mvc[i][1] = (int16_t)(((int)(int16_t)(mw[k] >> 16) * F.tscale + 128) >> 8);
                i++;
            }
    }
    return i;
}

/* lite: only what the reconstruction of a macroblock with known MVs needs (position, MV limits, source pixels) --
 * no neighbour types / motion, no skip prediction (second pass of a macroblock that is not P_SKIP) */
/* lite = 2: nothing of the neighbours AND no source pixels (the second pass of a macroblock that may turn out to need neither; it loads them itself when it re-encodes) */
PCAMV_DEV void mb_load(const FrameDev &F, MBLocal *L, int mb_x, int mb_y, int lite = 0, int rd = 0)
{
    L->mb_x = mb_x; L->mb_y = mb_y; L->mb_xy = mb_y * F.mb_w + mb_x;
    L->b_skip_mc = 0;
    MbFetch pf;
    if (!lite) {
    /* every load the macroblock needs is ISSUED before the first of them is used (types and motion of the neighbours here,
     * source pixels and the RD decision's neighbourhood in prim_mb_fetch): one memory round trip on the macroblock chain */
    const int top = L->mb_xy - F.mb_w;
    const int nb = (mb_y > 0 ? NB_TOP : 0) | (mb_x > 0 ? NB_LEFT : 0) | (mb_x < F.mb_w - 1 && mb_y > 0 ? NB_TOPRIGHT : 0) | (mb_x > 0 && mb_y > 0 ? NB_TOPLEFT : 0);
    /* (unconditional loads: a neighbour that does not exist reads this macroblock's own slot and is ignored -- a load inside an `if`
     * is a branch of its own with its own s_waitcnt, i.e. one memory round trip per neighbour instead of one for all) */
    int t_top = NB_LD8(&F.mb_type[(nb & NB_TOP) ? top : L->mb_xy]), t_left = NB_LD8(&F.mb_type[(nb & NB_LEFT) ? L->mb_xy - 1 : L->mb_xy]);
    int t_tr = NB_LD8(&F.mb_type[(nb & NB_TOPRIGHT) ? top + 1 : L->mb_xy]), t_tl = NB_LD8(&F.mb_type[(nb & NB_TOPLEFT) ? top - 1 : L->mb_xy]);
    const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w;
    const int b4 = 4 * (mb_y * s4 + mb_x), b8 = 2 * (mb_y * s8 + mb_x);
    const int t4 = (4 * (mb_y - 1) + 3) * s4 + 4 * mb_x, t8 = (2 * (mb_y - 1) + 1) * s8 + 2 * mb_x;
    /* the ten neighbouring 4x4 motion entries, one per lane: 0 top-left, 1..4 top, 5 top-right, 6..9 left */
    uint32_t nb_w[10]; int nb_r[10], nb_c8[10];       /* (one element per lane on the GPU: FOR_CAND's body runs once per lane) */
    FOR_CAND(i, 10) {
        int ok, c8, m4, r8;
        if (i == 0) { ok = nb & NB_TOPLEFT; c8 = SCAN8_0 - 1 - 8; m4 = t4 - 1; r8 = t8 - 1; }
        else if (i <= 4) { ok = nb & NB_TOP; c8 = SCAN8_0 - 8 + (i - 1); m4 = t4 + (i - 1); r8 = t8 + ((i - 1) >> 1); }
        else if (i == 5) { ok = nb & NB_TOPRIGHT; c8 = SCAN8_0 + 4 - 8; m4 = t4 + 4; r8 = t8 + 2; }
        else { ok = nb & NB_LEFT; c8 = SCAN8_0 - 1 + 8 * (i - 6); m4 = b4 - 1 + (i - 6) * s4; r8 = b8 - 1 + ((i - 6) >> 1) * s8; }
        nb_c8[NB_SLOT(i)] = ok ? c8 : -1;
        nb_w[NB_SLOT(i)] = NB_LD32(&F.mv[2 * (ok ? m4 : b4)]); nb_r[NB_SLOT(i)] = NB_LD8(&F.ref8[ok ? r8 : b8]);
    }
    prim_mb_fetch(F, mb_x, mb_y, nb, rd, pf);
    /* ---- from here on the loaded values are used */
    if (!(nb & NB_TOP)) t_top = -1;
    if (!(nb & NB_LEFT)) t_left = -1;
    if (!(nb & NB_TOPRIGHT)) t_tr = -1;
    if (!(nb & NB_TOPLEFT)) t_tl = -1;
    L->neighbour = nb;
    L->type_top = t_top; L->type_left = t_left; L->type_topright = t_tr; L->type_topleft = t_tl;
    PCAMV_WAVE_SYNC();
    FOR_CAND(i, 48) { L->cref[i] = -2; L->cmv[i][0] = 0; L->cmv[i][1] = 0; }
    PCAMV_WAVE_SYNC();
    FOR_CAND(i, 10) {
        const int c8 = nb_c8[NB_SLOT(i)];
        if (c8 >= 0) { const uint32_t w_ = nb_w[NB_SLOT(i)]; L->cref[c8] = (int8_t)nb_r[NB_SLOT(i)]; L->cmv[c8][0] = (int16_t)(w_ & 0xffff); L->cmv[c8][1] = (int16_t)(w_ >> 16); }
    }
    PCAMV_WAVE_SYNC();
    int pm[2];
    predict_mv_pskip(L, pm);
    L->pskip_mv[0] = (int16_t)pm[0]; L->pskip_mv[1] = (int16_t)pm[1];
    } else {
    L->neighbour = 0;
    L->type_left = L->type_top = L->type_topleft = L->type_topright = -1;
    }

    int fmv = 4 * F.mv_range;
    L->mv_min[0] = 4 * (-16 * mb_x - 24);
    L->mv_max[0] = 4 * (16 * (F.mb_w - mb_x - 1) + 24);
    L->mv_min_spel[0] = clip3i(L->mv_min[0], -fmv, fmv - 1);
    L->mv_max_spel[0] = clip3i(L->mv_max[0], -fmv, fmv - 1);
    L->mv_min_fpel[0] = (L->mv_min_spel[0] >> 2) + 5;
    L->mv_max_fpel[0] = (L->mv_max_spel[0] >> 2) - 5;
    L->mv_min[1] = 4 * (-16 * mb_y - 24);
    L->mv_max[1] = 4 * (16 * (F.mb_h - mb_y - 1) + 24);
    L->mv_min_spel[1] = clip3i(L->mv_min[1], imax(4 * (-512 + 8), -fmv), fmv);
    L->mv_max_spel[1] = clip3i(L->mv_max[1], -fmv, fmv - 1);
    L->mv_max_spel[1] = imin(L->mv_max_spel[1], fmv * 4);
    L->mv_min_fpel[1] = (L->mv_min_spel[1] >> 2) + 5;
    L->mv_max_fpel[1] = (L->mv_max_spel[1] >> 2) - 5;
    if (lite == 2) { }
    else if (lite) prim_load_fenc(F, L);
    else prim_mb_fetch_store(F, L, rd, pf);         /* source pixels; --subme >= 6: intra neighbours, entropy-coder neighbourhood, context states */
}

/* ---------------------------------------------------------------- motion search */
/* me.c / analyse.c tables as packed constants (see pcamv_common.h) */
/* subpel_iterations[subme] = {hpel0, qpel0, hpel1, qpel1}: {0,0,0,0},{1,1,0,0},{0,1,1,0},{0,2,1,0},{0,2,1,1},{0,2,1,2},{0,0,2,2},{0,0,2,2},{0,0,4,10},{0,0,4,10} */
PCAMV_DEV int subpel_iter_of(int subme, int k)
{
    if (subme < 4) return nib64(0x0120011000110000ull, 4 * subme + k);
    if (subme < 8) return nib64(0x2200220021201120ull, 4 * (subme - 4) + k);
    return nib32(0xa400a400u, 4 * (subme - 8) + k);
}
PCAMV_DEV int mod6m1_of(int i) { return nib32(0x05432105u, i); }                                   /* {5,0,1,2,3,4,5,0} */
PCAMV_DEV int hex2_x(int i) { return nib32(0x01343101u, i) - 2; }                                  /* {-1,-2,-1,1,2,1,-1,-2} */
PCAMV_DEV int hex2_y(int i) { return nib32(0x20024420u, i) - 2; }                                  /* {-2,0,2,2,0,-2,-2,0} */
PCAMV_DEV int hex4_x(int j) { return nib64(0x6422468888800000ull, j) - 4; }                        /* {-4,-4,-4,-4,-4,4,4,4,4,4,2,0,-2,-2,0,2} */
PCAMV_DEV int hex4_y(int j) { return nib64(0x1017876543223456ull, j) - 4; }                        /* {2,1,0,-1,-2,-2,-1,0,1,2,3,4,3,-3,-4,-3} */
PCAMV_DEV int range_mul_of(int mvd_ctx, int sad_ctx) { return nib64(0x6544544444434433ull, 4 * mvd_ctx + sad_ctx); } /* {3,3,4,4},{3,4,4,4},{4,4,4,5},{4,4,5,6} */
PCAMV_DEV int size_shift_of(int ip) { return nib32(0x4332110u, ip); }                              /* {0,1,1,2,3,3,4} */

#define MVCOSTX(v) prim_mv_cost(F, (v) - me->mvp[0])        /* one (wave-uniform) entry of the MV-bit table */
#define MVCOSTY(v) prim_mv_cost(F, (v) - me->mvp[1])

PCAMV_DEV EvalRes eval_cands(const FrameDev &F, MBLocal *L, MEState *me, const uint8_t *enc, int n, int flags)
{
    return prim_eval_list(F, L, enc, me->i_pixel, me->xoff, me->yoff, n, flags, me->mvp[0], me->mvp[1]);
}
/* The functions of the search are templates on TESA: 1 = the instance that can run --me tesa (its own kernel), 0 = the
 * instance every other method runs, with that method's code and -- more important -- its run-time choice of the
 * full-pel metric compiled out (the primitives are specialised on constant flags; with the flags a run-time value the
 * other methods lost 11 %).
 * encoder.c mbcmp_init: with --me tesa and subme > 1 the "full-pel" comparisons of the search (fpelcmp: COST_MV,
 * COST_MV_HPEL, the half-pel rounds of refine_subpel) are SATD instead of SAD.  FPEL_LIST: flags of a list of full-pel
 * candidates, FPEL_SAD: flags of a quarter-pel list scored with that metric. */
/* The template parameter is a variant mask: bit 0 = the --me tesa instance, bit 1 = the instance with the RD mode decision of
 * --subme >= 6 compiled in (a kernel of its own as well: its code would otherwise cost the search kernel registers). */
#define FPEL_SATD ((TESA & 1) && F.me_method == PCAMV_ME_TESA && F.subme > 1)
#define MBRD_ON ((TESA & 2) && F.b_mbrd)
/* bit 3: the instance that prices sub-8x8 partitions with x264_rd_cost_part (a build of its own: compiled into the others its code costs
 * them registers -- 22 spilled VGPRs and 135 more parked scalars in the 4-waves-per-SIMD build) */
#define RD_PSUB_ON ((TESA & 8) && (F.inter & PCAMV_ANALYSE_PSUB8x8))
#define FPEL_LIST (FPEL_SATD ? EV_SATD : EV_FPEL)
#define FPEL_SAD (FPEL_SATD ? EV_SATD : 0)
/* the n listed full-pel candidates folded into the running best, in list order (strict <) */
template <int TESA>
PCAMV_DEV int fpel_fold(const FrameDev &F, MBLocal *L, MEState *me, int &bmx, int &bmy, int &bcost, int n)
{
    EvalRes r = eval_cands(F, L, me, L->fenc, n, FPEL_LIST);
    if (r.cost < bcost) { bcost = r.cost; bmx = CAND_X(r.idx) >> 2; bmy = CAND_Y(r.idx) >> 2; return r.idx; }
    return -1;
}
#define FSET(c, X, Y) (L->cxy[c] = CAND_PACK((X) * 4, (Y) * 4))
#define TRY4(ox, oy, a0, a1, b0, b1, c0, c1, d0, d1) { \
        FSET(0, (ox) + (a0), (oy) + (a1)); FSET(1, (ox) + (b0), (oy) + (b1)); FSET(2, (ox) + (c0), (oy) + (c1)); FSET(3, (ox) + (d0), (oy) + (d1)); \
        fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, 4); }
#define TRY8(ox, oy, a0, a1, b0, b1, c0, c1, d0, d1, e0, e1, f0, f1, g0, g1, h0, h1) { \
        FSET(0, (ox) + (a0), (oy) + (a1)); FSET(1, (ox) + (b0), (oy) + (b1)); FSET(2, (ox) + (c0), (oy) + (c1)); FSET(3, (ox) + (d0), (oy) + (d1)); \
        FSET(4, (ox) + (e0), (oy) + (e1)); FSET(5, (ox) + (f0), (oy) + (f1)); FSET(6, (ox) + (g0), (oy) + (g1)); FSET(7, (ox) + (h0), (oy) + (h1)); \
        fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, 8); }
#define CHECK_MVRANGE(mx, my) ((mx) >= mv_x_min && (mx) <= mv_x_max && (my) >= mv_y_min && (my) <= mv_y_max)

template <int TESA>
PCAMV_DEV void refine_subpel(const FrameDev &F, MBLocal *L, MEState *me, int hpel_iters, int qpel_iters, int b_refine_qpel)
{
    const int ip = me->i_pixel;
    const int b_chroma_me = F.b_chroma_me && ip <= PIX_8x8;
    const int qflags = (F.subme > 1 ? EV_SATD : 0) | (b_chroma_me ? EV_CHROMA : 0);
    int bmx = me->mv[0], bmy = me->mv[1], bcost = me->cost, odir = -1, bdir;
    if (hpel_iters && F.subme < 3) {
        int mx = clip3i(me->mvp[0], L->mv_min_spel[0], L->mv_max_spel[0]);
        int my = clip3i(me->mvp[1], L->mv_min_spel[1], L->mv_max_spel[1]);
        if ((mx - bmx) | (my - bmy)) {
            L->cxy[0] = CAND_PACK(mx, my);
            EvalRes r = eval_cands(F, L, me, L->fenc, 1, FPEL_SAD);
            if (r.cost < bcost) { bcost = r.cost; bmx = mx; bmy = my; }
        }
    }
    for (int i = hpel_iters; i > 0; i--) {
        int omx = bmx, omy = bmy;
        L->cxy[0] = CAND_PACK(omx, omy - 2); L->cxy[1] = CAND_PACK(omx, omy + 2);
        L->cxy[2] = CAND_PACK(omx - 2, omy); L->cxy[3] = CAND_PACK(omx + 2, omy);
        EvalRes r = eval_cands(F, L, me, L->fenc, 4, FPEL_SAD);
        if (r.cost < bcost) { bcost = r.cost; bmx = CAND_X(r.idx); bmy = CAND_Y(r.idx); }
        if (bmx == omx && bmy == omy) break;
    }
    bdir = -1;
    int first_done = 0;
    if (!b_refine_qpel) {
        /* COST_MV_SATD of the half-pel result; the conditional chroma terms of the reference only skip
         * work for candidates that cannot win, so the full cost decides identically */
        if (bmy > L->mv_max_spel[1]) bmy = L->mv_max_spel[1];
        L->cxy[0] = CAND_PACK(bmx, bmy);
        if (qpel_iters > 0) {
            /* ... together with the first quarter-pel round around it (its four positions are known
             * already): candidate 0 sets the cost to beat, 1..4 are folded in order */
            const int omx = bmx, omy = bmy;
            L->cxy[1] = CAND_PACK(omx, omy - 1); L->cxy[2] = CAND_PACK(omx, omy + 1);
            L->cxy[3] = CAND_PACK(omx - 1, omy); L->cxy[4] = CAND_PACK(omx + 1, omy);
            eval_cands(F, L, me, L->fenc, 5, qflags);
            bcost = L->ccost[0];
            for (int k = 1; k < 5; k++)
                if (L->ccost[k] < bcost) { bcost = L->ccost[k]; bmx = CAND_X(k); bmy = CAND_Y(k); bdir = k - 1; }
            first_done = (bmx == omx && bmy == omy) ? 2 : 1;       /* 2: no move, the round loop ends here */
        } else
            bcost = eval_cands(F, L, me, L->fenc, 1, qflags).cost;
    }
    for (int i = qpel_iters - (first_done ? 1 : 0); i > 0 && first_done != 2; i--) {
        odir = bdir;
        int omx = bmx, omy = bmy;
        L->cxy[0] = CAND_PACK(omx, omy - 1); L->cxy[1] = CAND_PACK(omx, omy + 1);
        L->cxy[2] = CAND_PACK(omx - 1, omy); L->cxy[3] = CAND_PACK(omx + 1, omy);
        if (!b_refine_qpel && odir >= 0) L->cxy[odir ^ 1] = CAND_NONE;      /* never step straight back */
        EvalRes r = eval_cands(F, L, me, L->fenc, 4, qflags);
        if (r.cost < bcost) { bcost = r.cost; bmx = CAND_X(r.idx); bmy = CAND_Y(r.idx); bdir = r.idx; }
        if (bmx == omx && bmy == omy) break;
    }
    if (bmy > L->mv_max_spel[1]) {
        bmy = L->mv_max_spel[1];
        L->cxy[0] = CAND_PACK(bmx, bmy);
        bcost = eval_cands(F, L, me, L->fenc, 1, qflags).cost;
    }
    me->cost = bcost; me->mv[0] = bmx; me->mv[1] = bmy;
    me->cost_mv = MVCOSTX(bmx) + MVCOSTY(bmy);
}

/* the two arms of the UMH cross (me.c:331-357 CROSS): +i, -i for i = start, start+2, .. < max, first
 * along x then along y; candidates beyond the search window are skipped */
template <int TESA>
PCAMV_DEV void cross_search(const FrameDev &F, MBLocal *L, MEState *me, int &bmx, int &bmy, int &bcost,
                            int omx, int omy, int start, int x_max, int y_max,
                            int mv_x_min, int mv_x_max, int mv_y_min, int mv_y_max)
{
    const int nx = x_max > start ? (x_max - start + 1) >> 1 : 0;
    const int ny = y_max > start ? (y_max - start + 1) >> 1 : 0;
    const int total = 2 * (nx + ny);
    for (int base = 0; base < total; base += 64) {
        const int n = imin(64, total - base);
        FOR_CAND(c, n) {
            int g = base + c, isy = g >= 2 * nx, hh = isy ? g - 2 * nx : g;
            int i = start + 2 * (hh >> 1), neg = hh & 1, d = neg ? -i : i;
            int x = isy ? omx : omx + d, y = isy ? omy + d : omy;
            int ok = isy ? (neg ? y >= mv_y_min : y <= mv_y_max) : (neg ? x >= mv_x_min : x <= mv_x_max);
            L->cxy[c] = ok ? CAND_PACK(x * 4, y * 4) : CAND_NONE;
        }
        fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, n);
    }
}

template <int TESA>
PCAMV_DEV void hex_search(const FrameDev &F, MBLocal *L, MEState *me, int &bmx, int &bmy, int &bcost, int i_me_range,
                          int mv_x_min, int mv_x_max, int mv_y_min, int mv_y_max)
{
    int dir = -2;
    FOR_CAND(c, 6) L->cxy[c] = CAND_PACK((bmx + hex2_x(c + 1)) * 4, (bmy + hex2_y(c + 1)) * 4);
    {
        EvalRes r = eval_cands(F, L, me, L->fenc, 6, FPEL_LIST);
        if (r.cost < bcost) { bcost = r.cost; dir = r.idx; }
    }
    if (dir != -2) {
        bmx += hex2_x(dir + 1); bmy += hex2_y(dir + 1);
        for (int i = 1; i < i_me_range / 2 && CHECK_MVRANGE(bmx, bmy); i++) {
            const int odir = mod6m1_of(dir + 1);
            FOR_CAND(c, 3) L->cxy[c] = CAND_PACK((bmx + hex2_x(odir + c)) * 4, (bmy + hex2_y(odir + c)) * 4);
            EvalRes r = eval_cands(F, L, me, L->fenc, 3, FPEL_LIST);
            dir = -2;
            if (r.cost < bcost) { bcost = r.cost; dir = odir - 1 + r.idx; }
            if (dir == -2) break;
            bmx += hex2_x(dir + 1); bmy += hex2_y(dir + 1);
        }
    }
    int omx = bmx, omy = bmy;
    TRY8(omx, omy, 0, -1, 0, 1, -1, 0, 1, 0, -1, -1, -1, 1, 1, -1, 1, 1);
}

/* Hadamard exhaustive search (me.c:525-600): the window of ESA, but a position is only remembered when its ADS and
 * then its SAD pass thresholds relative to the best SAD so far; the list is pruned to me_range / 2 entries and those
 * are scored with the search's comparison function (SATD above subme 1).  */
template <int TESA>
PCAMV_DEV void tesa_search(const FrameDev &F, MBLocal *L, MEState *me, int &bmx, int &bmy, int &bcost, int i_me_range,
                                      int mv_x_min, int mv_x_max, int mv_y_min, int mv_y_max)
{
    const int ip = me->i_pixel;
    const int min_x = imax(bmx - i_me_range, mv_x_min), min_y = imax(bmy - i_me_range, mv_y_min);
    const int max_x = imin(bmx + i_me_range, mv_x_max), max_y = imin(bmy + i_me_range, mv_y_max);
    const int width = (max_x - min_x + 3) & ~3;
    if (width > 0 && max_y >= min_y) {
        const int sad_thresh = i_me_range <= 16 ? 10 : i_me_range <= 24 ? 11 : 12;
        int n = 0;
        FSET(0, bmx, bmy);
        int bsad = eval_cands(F, L, me, L->fenc, 1, EV_FPEL).cost;
        for (int my = min_y; my <= max_y; my++) {
            const int ycost = MVCOSTY(my * 4);
            if (bsad <= ycost) continue;
            bsad -= ycost;
            prim_tesa_row(F, L, ip, me->xoff, me->yoff, min_x, my, width, me->mvp[0]);
            bsad = prim_tesa_scan(L, width, bsad, sad_thresh, ycost, my - min_y, &n);
            bsad += ycost;
        }
        n = prim_tesa_select(L, n, i_me_range / 2, bsad, sad_thresh, min_x, min_y);
        if (n > 0) fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, n);
    }
}

template <int TESA>
PCAMV_DEV void me_search_body(const FrameDev &F, MBLocal *L, MEState *me, int (*mvc)[2], int i_mvc);
#if defined(PCAMV_SEARCH_CALL) && !defined(PCAMV_HOST_EMU)
/* -DPCAMV_SEARCH_CALL (off; kept for measurements): the search of one partition as a REAL function -- 5 to 9 calls per
 * macroblock, inside which the register file is the search's alone; the arguments go through LDS, the frame descriptor is read
 * again from constant memory (a reference to the caller's copy would pin that copy in scratch).  Measured in the
 * 4-waves-per-SIMD build of the RD kernel: 13.7 against 14.4 M MB/s at 4096 chains -- the callee spills as much as the inlined
 * code did (195 against 149 registers over caller + callee), and the descriptor reload and the LDS hand-over are new. */
template <int TESA>
static __device__ __noinline__ void me_search_fn(MBLocal *L, int i_mvc_)
{
    const unsigned long long fp = (unsigned long long)L->fdesc;
    const __attribute__((address_space(4))) unsigned *src = (const __attribute__((address_space(4))) unsigned *)
        (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(fp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)fp));
    FrameDev Fm;
    unsigned *dst = (unsigned *)&Fm;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FrameDev) / 4); i++) dst[i] = src[i];
    const FrameDev &F = Fm;
    me_search_body<TESA>(F, L, (MEState *)L->me_tmp, L->mvc_tmp, __builtin_amdgcn_readfirstlane(i_mvc_));
}
template <int TESA>
PCAMV_DEV void me_search(const FrameDev &F, MBLocal *L, MEState *me, int (*mvc)[2], int i_mvc)
{
    (void)F;
    PCAMV_WAVE_SYNC();
    *(MEState *)L->me_tmp = *me;
    for (int i = 0; i < i_mvc; i++) { L->mvc_tmp[i][0] = mvc[i][0]; L->mvc_tmp[i][1] = mvc[i][1]; }
    PCAMV_WAVE_SYNC();
    me_search_fn<TESA>(L, i_mvc);
    PCAMV_WAVE_SYNC();
    *me = *(MEState *)L->me_tmp;
}
#else
template <int TESA>
PCAMV_DEV void me_search(const FrameDev &F, MBLocal *L, MEState *me, int (*mvc)[2], int i_mvc) { me_search_body<TESA>(F, L, me, mvc, i_mvc); }
#endif
template <int TESA>
PCAMV_DEV void me_search_body(const FrameDev &F, MBLocal *L, MEState *me, int (*mvc)[2], int i_mvc)
{
    const int ip = me->i_pixel;
    int i_me_range = F.me_range;
    int bmx, bmy, bcost, bpred_mx = 0, bpred_my = 0, bpred_cost = PCAMV_COST_MAX, omx, omy, pmx, pmy;
    int umh_ucost1 = 0, umh_diamonds_done = 0;
    const int mv_x_min = L->mv_min_fpel[0], mv_y_min = L->mv_min_fpel[1], mv_x_max = L->mv_max_fpel[0], mv_y_max = L->mv_max_fpel[1];

    bmx = clip3i(me->mvp[0], mv_x_min * 4, mv_x_max * 4);
    bmy = clip3i(me->mvp[1], mv_y_min * 4, mv_y_max * 4);
    pmx = (bmx + 2) >> 2; pmy = (bmy + 2) >> 2;
    bcost = PCAMV_COST_MAX;

    if (F.subme >= 3) {
        /* quarter-pel test of the predictor and the candidates (plain SAD), me.c:202-230 */
        int n = 1, sx = bmx, sy = bmy;
        L->cxy[0] = CAND_PACK(bmx, bmy);
        for (int i = 0; i < i_mvc; i++)
            if ((mvc[i][0] | mvc[i][1]) && ((sx - mvc[i][0]) | (sy - mvc[i][1]))) {
                L->cxy[n] = CAND_PACK(clip3i(mvc[i][0], mv_x_min * 4, mv_x_max * 4), clip3i(mvc[i][1], mv_y_min * 4, mv_y_max * 4));
                n++;
            }
        EvalRes r = eval_cands(F, L, me, L->fenc, n, FPEL_SAD);
        bpred_cost = r.cost; bpred_mx = CAND_X(r.idx); bpred_my = CAND_Y(r.idx);
        bmx = (bpred_mx + 2) >> 2; bmy = (bpred_my + 2) >> 2;
        FSET(0, bmx, bmy); FSET(1, 0, 0);
        if (F.me_method == PCAMV_ME_UMH) {
            /* UMH goes on with the small diamonds around the predictor and around (0,0) (me.c:308-316),
             * whose positions do not depend on the two tests above: one list, same order */
            int n = 6;
            FSET(2, pmx, pmy - 1); FSET(3, pmx, pmy + 1); FSET(4, pmx - 1, pmy); FSET(5, pmx + 1, pmy);
            if (pmx | pmy) { FSET(6, 0, -1); FSET(7, 0, 1); FSET(8, -1, 0); FSET(9, 1, 0); n = 10; }
            fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, n);
            umh_ucost1 = imin(L->ccost[0], L->ccost[1]);
            umh_diamonds_done = 1;
        } else
            fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, 2);
    } else {
        /* full-pel test: the predictor (its MV bits not charged), then the candidates, then (0,0).
         * Candidates equal to the running best are listed too: their cost cannot be smaller. */
        int n = 1;
        FSET(0, pmx, pmy);
        for (int i = 0; i < i_mvc; i++) {
            int mx = (mvc[i][0] + 2) >> 2, my = (mvc[i][1] + 2) >> 2;
            if (mx | my) { FSET(n, clip3i(mx, mv_x_min, mv_x_max), clip3i(my, mv_y_min, mv_y_max)); n++; }
        }
        FSET(n, 0, 0); n++;
        eval_cands(F, L, me, L->fenc, n, FPEL_LIST);
        bcost = L->ccost[0] - (MVCOSTX(pmx * 4) + MVCOSTY(pmy * 4)); bmx = pmx; bmy = pmy;
        for (int k = 1; k < n; k++)
            if (L->ccost[k] < bcost) { bcost = L->ccost[k]; bmx = CAND_X(k) >> 2; bmy = CAND_Y(k) >> 2; }
    }

    if (F.me_method == PCAMV_ME_DIA) {
        int i = 0;
        do {
            omx = bmx; omy = bmy;
            TRY4(omx, omy, 0, -1, 0, 1, -1, 0, 1, 0);
            if ((bmx == omx) & (bmy == omy)) break;
            if (!CHECK_MVRANGE(bmx, bmy)) break;
        } while (++i < i_me_range);
    } else if (F.me_method == PCAMV_ME_HEX) {
        hex_search<TESA>(F, L, me, bmx, bmy, bcost, i_me_range, mv_x_min, mv_x_max, mv_y_min, mv_y_max);
    } else if (F.me_method == PCAMV_ME_ESA) {
        /* exhaustive search (me.c:489-622).  The reference walks the window row by row, drops
         * positions whose sum-of-block-DC difference (ADS, a lower bound of the SAD) plus MV bits
         * cannot beat the running best, and tests the rest in raster order with strict <.  A dropped
         * position could never have won, so the first minimum over ALL positions of the same window
         * in the same order is the same result; no integral image is needed here.  The window is the
         * reference's: width rounded as (max_x - min_x + 3) & ~3 columns starting at min_x. */
        const int min_x = imax(bmx - i_me_range, mv_x_min), min_y = imax(bmy - i_me_range, mv_y_min);
        const int max_x = imin(bmx + i_me_range, mv_x_max), max_y = imin(bmy + i_me_range, mv_y_max);
        const int width = (max_x - min_x + 3) & ~3;
        if (width > 0 && max_y >= min_y) {
            EvalRes r = prim_esa_window(F, L, ip, me->xoff, me->yoff, min_x, min_y, width, max_y - min_y + 1, me->mvp[0], me->mvp[1]);
            if (r.cost < bcost) { bcost = r.cost; bmx = min_x + r.idx % width; bmy = min_y + r.idx / width; }
        }
    } else if ((TESA & 1) && F.me_method == PCAMV_ME_TESA) {
        tesa_search<TESA>(F, L, me, bmx, bmy, bcost, i_me_range, mv_x_min, mv_x_max, mv_y_min, mv_y_max);
    } else { /* UMH */
        int ucost1, ucost2, cross_start = 1, do_hex = 1, done = 0;
#define SAD_THRESH(v) (bcost < ((v) >> size_shift_of(ip)))
        if (umh_diamonds_done) ucost1 = umh_ucost1;
        else {
            ucost1 = bcost;
            if (pmx | pmy) { TRY8(0, 0, pmx, pmy - 1, pmx, pmy + 1, pmx - 1, pmy, pmx + 1, pmy, 0, -1, 0, 1, -1, 0, 1, 0); }
            else { TRY4(pmx, pmy, 0, -1, 0, 1, -1, 0, 1, 0); }
        }
        if (ip != PIX_4x4) {
            ucost2 = bcost;
            if ((bmx | bmy) && ((bmx - pmx) | (bmy - pmy))) { omx = bmx; omy = bmy; TRY4(omx, omy, 0, -1, 0, 1, -1, 0, 1, 0); }
            if (bcost == ucost2) cross_start = 3;
            omx = bmx; omy = bmy;
            if (bcost == ucost2 && SAD_THRESH(2000)) {
                TRY8(omx, omy, 0, -2, -1, -1, 1, -1, -2, 0, 2, 0, -1, 1, 1, 1, 0, 2);
                if (bcost == ucost1 && SAD_THRESH(500)) { done = 1; do_hex = 0; }
                else if (bcost == ucost2) {
                    int range = (i_me_range >> 1) | 1;
                    cross_search<TESA>(F, L, me, bmx, bmy, bcost, omx, omy, 3, range, range, mv_x_min, mv_x_max, mv_y_min, mv_y_max);
                    TRY8(omx, omy, -1, -2, 1, -2, -2, -1, 2, -1, -2, 1, 2, 1, -1, 2, 1, 2);
                    if (bcost == ucost2) { done = 1; do_hex = 0; }
                    cross_start = range + 2;
                }
            }
            if (!done) {
                if (i_mvc) {
                    int mvd, sad_ctx, mvd_ctx, denom = 1;
                    if (i_mvc == 1) {
                        if (ip == PIX_16x16) mvd = 25;
                        else mvd = iabs(me->mvp[0] - mvc[0][0]) + iabs(me->mvp[1] - mvc[0][1]);
                    } else {
                        denom = i_mvc - 1; mvd = 0;
                        if (ip != PIX_16x16) { mvd = iabs(me->mvp[0] - mvc[0][0]) + iabs(me->mvp[1] - mvc[0][1]); denom++; }
                        for (int i = 0; i < i_mvc - 1; i++) mvd += iabs(mvc[i][0] - mvc[i + 1][0]) + iabs(mvc[i][1] - mvc[i + 1][1]);
                    }
                    sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
                    mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                    i_me_range = i_me_range * range_mul_of(mvd_ctx, sad_ctx) / 4;
                }
                cross_search<TESA>(F, L, me, bmx, bmy, bcost, omx, omy, cross_start, i_me_range, i_me_range / 2, mv_x_min, mv_x_max, mv_y_min, mv_y_max);
                TRY4(omx, omy, -2, -2, -2, 2, 2, -2, 2, 2);
                /* 16-point hexagons at radii 4, 8, .. around the (fixed) cross result, me.c:404-457 */
                omx = bmx; omy = bmy;
                const int total = 16 * imax(1, i_me_range / 4);
                for (int base = 0; base < total; base += 64) {
                    const int n = imin(64, total - base);
                    FOR_CAND(c, n) {
                        int g = base + c, i = 1 + (g >> 4), j = g & 15;
                        int mx = omx + hex4_x(j) * i, my = omy + hex4_y(j) * i;
                        L->cxy[c] = CHECK_MVRANGE(mx, my) ? CAND_PACK(mx * 4, my * 4) : CAND_NONE;
                    }
                    fpel_fold<TESA>(F, L, me, bmx, bmy, bcost, n);
                }
                if (!(bmy <= mv_y_max)) do_hex = 0;
            }
        }
        if (do_hex) hex_search<TESA>(F, L, me, bmx, bmy, bcost, i_me_range, mv_x_min, mv_x_max, mv_y_min, mv_y_max);
#undef SAD_THRESH
    }

    if (bpred_cost < bcost) { me->mv[0] = bpred_mx; me->mv[1] = bpred_my; me->cost = bpred_cost; }
    else { me->mv[0] = bmx * 4; me->mv[1] = bmy * 4; me->cost = bcost; }
    me->cost_mv = MVCOSTX(me->mv[0]) + MVCOSTY(me->mv[1]);
    if (bmx == pmx && bmy == pmy && F.subme < 3) me->cost += me->cost_mv;
    if (F.subme >= 2) refine_subpel<TESA>(F, L, me, subpel_iter_of(F.subme, 2), subpel_iter_of(F.subme, 3), 0);
    else if (me->mv[1] > L->mv_max_spel[1]) me->mv[1] = L->mv_max_spel[1];
}

template <int TESA>
PCAMV_DEV void me_refine_qpel(const FrameDev &F, MBLocal *L, MEState *me)
{
    refine_subpel<TESA>(F, L, me, subpel_iter_of(F.subme, 0), subpel_iter_of(F.subme, 1), 1);
}

/* ---------------------------------------------------------------- macroblock (re-)encode */
/* sequential-equivalent decimation of a 4x4 scan: done per block by prim_residual (blk_score) */
/* lv: also leave the quantised levels in scan order (L->coef, L->cdc) and the per-block non-zero flags (L->nzc) for the
 * entropy coder's size / context walk (--subme >= 6) */
PCAMV_DEV void mb_encode(const FrameDev &F, MBLocal *L, int win = 0, int lv = 0)
{
    if (L->i_type == PCAMV_P_SKIP) {
        if (!L->b_skip_mc) {
            int mvx = clip3i(L->cmv[SCAN8_0][0], L->mv_min[0], L->mv_max[0]);
            int mvy = clip3i(L->cmv[SCAN8_0][1], L->mv_min[1], L->mv_max[1]);
            prim_predict_16x16(F, L, mvx, mvy, 1);
        }
        L->cbp_luma = L->cbp_chroma = 0;
        L->nnz_mask = 0;
        return;
    }
    if (!L->b_skip_mc) prim_predict_mb(F, L, win);
    prim_mb_transform(F, L, lv);
}

PCAMV_DEV int probe_pskip(const FrameDev &F, MBLocal *L)
{
    int mvx = clip3i(L->pskip_mv[0], L->mv_min[0], L->mv_max[0]);
    int mvy = clip3i(L->pskip_mv[1], L->mv_min[1], L->mv_max[1]);
    prim_predict_16x16(F, L, mvx, mvy, 0);
    prim_residual(F, L, 1, 0);
    int decimate = 0;
    for (int idx = 0; idx < 16; idx++) if (L->blk_nz[idx]) { decimate += L->blk_score[idx]; if (decimate >= 6) return 0; }
    int thresh = (F.lambda2_chroma + 32) >> 6;
    prim_predict_16x16(F, L, mvx, mvy, 2);
    int need[2];
    for (int ch = 0; ch < 2; ch++) need[ch] = !(prim_chroma_ssd(F, L, ch) < thresh);
    if (need[0] | need[1]) prim_residual(F, L, 0, 1);
    for (int ch = 0; ch < 2; ch++) {
        if (!need[ch]) continue;
        int mf = F.q_mf[1][0] >> 1, bias = F.q_bias[1][0] << 1, nz = 0;
        for (int k = 0; k < 4; k++) {
            int c = L->cdc[ch][k];
            nz |= c > 0 ? ((bias + c) * mf >> 16) : -((bias - c) * mf >> 16);
        }
        if (nz) return 0;
        decimate = 0;
        for (int i = 0; i < 4; i++) if (L->blk_nz[16 + ch * 4 + i]) { decimate += L->blk_score[16 + ch * 4 + i]; if (decimate >= 7) return 0; }
    }
    L->b_skip_mc = 1;
    return 1;
}

/* ---------------------------------------------------------------- partition analysis */
struct Analysis {
    MEState me16x16, me8x8[4], me16x8[2], me8x16[2];
    MEState me4x4[4][4], me8x4[4][2], me4x8[4][2];
    int mvc[5][2];
    int cost8x8, cost16x8, cost8x16, cost4x4[4], cost8x4[4], cost4x8[4];
    int rd16x16;
    int rd16_early;                /* the 16x16 search ended on the skip MV: its RD trial is made whatever the thresholds say (analyse.c:1194-1203) */
    int sel_type, sel_part, sel_cost;      /* the decision by SATD cost (analyse.c:2653-2743), input of the refinement / RD stage */
};
PCAMV_DEV void me_setup(MEState *me, int ip, int xoff, int yoff) { me->i_pixel = ip; me->xoff = xoff; me->yoff = yoff; me->cost = me->cost_mv = me->cost_rec = 0; me->mv[0] = me->mv[1] = 0; me->mvp[0] = me->mvp[1] = 0; }

PCAMV_DEV void cache_mv_p8x8(MBLocal *L, Analysis *a, int i)
{
    int x = 2 * (i % 2), y = 2 * (i / 2);
    switch (L->sub_part[i]) {
    case PCAMV_D_L0_8x8: cache_mv_set(L, x, y, 2, 2, a->me8x8[i].mv[0], a->me8x8[i].mv[1]); break;
    case PCAMV_D_L0_8x4: cache_mv_set(L, x, y, 2, 1, a->me8x4[i][0].mv[0], a->me8x4[i][0].mv[1]); cache_mv_set(L, x, y + 1, 2, 1, a->me8x4[i][1].mv[0], a->me8x4[i][1].mv[1]); break;
    case PCAMV_D_L0_4x8: cache_mv_set(L, x, y, 1, 2, a->me4x8[i][0].mv[0], a->me4x8[i][0].mv[1]); cache_mv_set(L, x + 1, y, 1, 2, a->me4x8[i][1].mv[0], a->me4x8[i][1].mv[1]); break;
    default:
        cache_mv_set(L, x, y, 1, 1, a->me4x4[i][0].mv[0], a->me4x4[i][0].mv[1]); cache_mv_set(L, x + 1, y, 1, 1, a->me4x4[i][1].mv[0], a->me4x4[i][1].mv[1]);
        cache_mv_set(L, x, y + 1, 1, 1, a->me4x4[i][2].mv[0], a->me4x4[i][2].mv[1]); cache_mv_set(L, x + 1, y + 1, 1, 1, a->me4x4[i][3].mv[0], a->me4x4[i][3].mv[1]); break;
    }
}
PCAMV_DEV void update_cache(MBLocal *L, Analysis *a)
{
    if (L->i_type == PCAMV_P_L0) {
        cache_ref_set(L, 0, 0, 4, 4, 0);
        if (L->i_partition == PCAMV_D_16x16) cache_mv_set(L, 0, 0, 4, 4, a->me16x16.mv[0], a->me16x16.mv[1]);
        else if (L->i_partition == PCAMV_D_16x8) { cache_mv_set(L, 0, 0, 4, 2, a->me16x8[0].mv[0], a->me16x8[0].mv[1]); cache_mv_set(L, 0, 2, 4, 2, a->me16x8[1].mv[0], a->me16x8[1].mv[1]); }
        else if (L->i_partition == PCAMV_D_8x16) { cache_mv_set(L, 0, 0, 2, 4, a->me8x16[0].mv[0], a->me8x16[0].mv[1]); cache_mv_set(L, 2, 0, 2, 4, a->me8x16[1].mv[0], a->me8x16[1].mv[1]); }
    } else if (L->i_type == PCAMV_P_8x8) {
        cache_ref_set(L, 0, 0, 4, 4, 0);
        for (int i = 0; i < 4; i++) cache_mv_p8x8(L, a, i);
    } else {
        L->i_partition = PCAMV_D_16x16;
        cache_ref_set(L, 0, 0, 4, 4, 0);
        cache_mv_set(L, 0, 0, 4, 4, L->pskip_mv[0], L->pskip_mv[1]);
    }
}

/* ---------------------------------------------------------------- --subme >= 6: RD mode decision (i_mbrd = 1, analyse.c:236)
 *   intra SATD analysis (only its cost enters, as a bound on the RD trials)   analyse.c:552-879
 *   size-only CABAC walk of the macroblock layer / context adaptation       rdo.c:49-62, encoder/cabac.c:85-113, 234-330, 403-470, 777-1018
 *   x264_rd_cost_mb, x264_mb_analyse_p_rd                                    rdo.c:139-171, analyse.c:2117-2186 */
PCAMV_DEV int intra_avail(int nb) { return (nb & NB_TOPLEFT) ? 3 : (nb & NB_LEFT) ? 1 : (nb & NB_TOP) ? 2 : 0; }      /* bit 0: left, bit 1: top (predict_*_mode_available) */
PCAMV_DEV int intra_chroma_cost(const FrameDev &F, MBLocal *L)
{
    prim_intra8c_satd(F, L, intra_avail(L->neighbour));      /* ccost[mode]: DC (its variant), H, V, P over both planes; COST_MAX = not available */
    int best = PCAMV_COST_MAX;
    for (int m = 0; m < 4; m++) {
        const int c = L->ccost[m];
        if (c < PCAMV_COST_MAX) best = imin(best, c + F.lambda * (m == 0 ? 1 : m == 3 ? 5 : 3));       /* lambda * bs_size_ue(mode) */
    }
    return best;
}
/* common/macroblock.c:765-774, 1226-1238: the neighbours 4x4 block i has */
PCAMV_DEV int intra4_neighbours(int nb, int i)
{
    if (i == 0) return (nb & (NB_TOP | NB_LEFT | NB_TOPLEFT)) | ((nb & NB_TOP) ? NB_TOPRIGHT : 0);
    if (i == 1 || i == 4) return NB_LEFT | ((nb & NB_TOP) ? (NB_TOP | NB_TOPLEFT | NB_TOPRIGHT) : 0);
    if (i == 2 || i == 8 || i == 10) return NB_TOP | NB_TOPRIGHT | ((nb & NB_LEFT) ? (NB_LEFT | NB_TOPLEFT) : 0);
    if (i == 5) return NB_LEFT | (nb & NB_TOPRIGHT) | ((nb & NB_TOP) ? (NB_TOP | NB_TOPLEFT) : 0);
    return NB_LEFT | NB_TOP | NB_TOPLEFT | ((i == 3 || i == 7 || i == 11 || i == 13 || i == 15) ? 0 : NB_TOPRIGHT);
}
PCAMV_DEV int i4_fix(int m) { return m < 0 ? -1 : m > I4_HU ? I4_DC : m; }
PCAMV_DEV void intra_analyse(const FrameDev &F, MBLocal *L, int i_satd_inter, int *i16, int *i4)
{
    *i16 = *i4 = PCAMV_COST_MAX;
    prim_intra16_satd(F, L, intra_avail(L->neighbour));      /* ccost[mode]: V, H, DC (its variant), P */
    for (int m = 0; m < 4; m++) {
        const int c = L->ccost[m];
        if (c < PCAMV_COST_MAX) *i16 = imin(*i16, c + F.lambda * (m == 0 ? 1 : m == 3 ? 5 : 3));
    }
    if (L->b_fast_intra && *i16 > 2 * i_satd_inter) return;
    if (!(F.inter & PCAMV_ANALYSE_I4x4)) return;
    int thresh = imin(i_satd_inter, *i16);
    thresh = thresh * (10 - L->b_fast_intra) / 8;
    int cost = F.lambda * 24, idx;
    prim_intra4_init(L);
    for (idx = 0;; idx++) {
        const int n4 = intra4_neighbours(L->neighbour, idx);
        const int fa = i4_fix(L->i4mode[scan8_of(idx) - 1]), fb = i4_fix(L->i4mode[scan8_of(idx) - 8]);
        const int pred_mode = imin(fa, fb) < 0 ? I4_DC : imin(fa, fb);
        int n = 0;
        PCAMV_WAVE_SYNC();
        if ((n4 & NB_LEFT) && (n4 & NB_TOP)) {
            L->slots[n++] = I4_DC; L->slots[n++] = I4_H; L->slots[n++] = I4_V; L->slots[n++] = I4_DDL;
            if (n4 & NB_TOPLEFT) { L->slots[n++] = I4_DDR; L->slots[n++] = I4_VR; L->slots[n++] = I4_HD; }
            L->slots[n++] = I4_VL; L->slots[n++] = I4_HU;
        } else if (n4 & NB_LEFT) { L->slots[n++] = I4_DC_LEFT; L->slots[n++] = I4_H; L->slots[n++] = I4_HU; }
        else if (n4 & NB_TOP) { L->slots[n++] = I4_DC_TOP; L->slots[n++] = I4_V; L->slots[n++] = I4_DDL; L->slots[n++] = I4_VL; }
        else L->slots[n++] = I4_DC_128;
        prim_intra4_costs(F, L, idx, n, (n4 & (NB_TOPRIGHT | NB_TOP)) == NB_TOP);     /* ccost[i] = SATD of mode slots[i] */
        int best = PCAMV_COST_MAX, best_mode = 0;
        for (int i = 0; i < n; i++) {
            const int c = L->ccost[i] + F.lambda * (pred_mode == i4_fix(L->slots[i]) ? 1 : 4);
            if (c < best) { best = c; best_mode = L->slots[i]; }
        }
        cost += best;
        if (cost > thresh || idx == 15) break;
        prim_intra4_encode(F, L, idx, best_mode);
        L->i4mode[scan8_of(idx)] = (int8_t)best_mode;
    }
    if (idx == 15) *i4 = cost;
}

/* The CABAC walk of a macroblock goes through a CabWalk (primitives: on the GPU every context's state lives in the lane that
 * owns it while the macroblock is walked; pcamv_prims_rd_gpu.h): prim_cab_begin, prim_cb_dec (one decision, context and bin
 * wave-uniform), prim_cb_bypass, prim_cab_residual, prim_cab_end. */
PCAMV_DEV void cb_mvd_cpn(MBLocal *L, CabWalk &C, int idx, int l, int mvd)      /* encoder/cabac.c:403-449 */
{
    const int amvd = iabs(L->cmvd[scan8_of(idx) - 1][l]) + iabs(L->cmvd[scan8_of(idx) - 8][l]);
    const int a = iabs(mvd), base = l ? 47 : 40;
    prim_cb_dec(L, C, base + (amvd > 2) + (amvd > 32), a != 0);
    if (!a) return;
    for (int i = 1; i < imin(a, 4); i++) prim_cb_dec(L, C, base + i + 2, 1);               /* contexts 3, 4, 5, then 6, 6, .. */
    if (a >= 4) prim_cb_run(L, C, base + 6, imin(a, 9) - 4, a < 9);                        /* (the decisions on context 6 as one run) */
    else prim_cb_dec(L, C, base + a + 2, 0);
    if (a >= 9) prim_cb_bypass(C, (size_ue_of((unsigned)(a - 9 + 7)) - 3) << 8);            /* Exp-Golomb k = 3 suffix */
    prim_cb_bypass(C, 256);                                                                /* sign */
}
PCAMV_DEV void cb_mvd(MBLocal *L, CabWalk &C, int idx, int width, int height)     /* encoder/cabac.c:452-470 */
{
    int mvp[2];
    predict_mv(L, idx, width, mvp);
    const int dx = L->cmv[scan8_of(idx)][0] - mvp[0], dy = L->cmv[scan8_of(idx)][1] - mvp[1];
    cb_mvd_cpn(L, C, idx, 0, dx);
    cb_mvd_cpn(L, C, idx, 1, dy);
    for (int j = 0; j < height; j++)
        for (int i = 0; i < width; i++) { L->cmvd[scan8_of(idx) + i + 8 * j][0] = (int16_t)dx; L->cmvd[scan8_of(idx) + i + 8 * j][1] = (int16_t)dy; }
}
/* macroblock layer of a P_L0 / P_8x8 macroblock up to the residual: mb_type, sub_mb_type, mvd, coded_block_pattern,
 * mb_qp_delta (0: constant QP, last delta 0) */
PCAMV_DEV void cabac_mb_header(const FrameDev &F, MBLocal *L, CabWalk &C)
{
    prim_cb_dec(L, C, 14, 0);
    if (L->i_type == PCAMV_P_8x8 || L->i_partition == PCAMV_D_16x16) { prim_cb_dec(L, C, 15, 0); prim_cb_dec(L, C, 16, L->i_type == PCAMV_P_8x8); }
    else { prim_cb_dec(L, C, 15, 1); prim_cb_dec(L, C, 17, L->i_partition == PCAMV_D_16x8); }
    if (L->i_type == PCAMV_P_8x8) {
        for (int i = 0; i < 4; i++) {
            const int sp = L->sub_part[i];
            prim_cb_dec(L, C, 21, sp == PCAMV_D_L0_8x8);
            if (sp != PCAMV_D_L0_8x8) { prim_cb_dec(L, C, 22, sp != PCAMV_D_L0_8x4); if (sp != PCAMV_D_L0_8x4) prim_cb_dec(L, C, 23, sp == PCAMV_D_L0_4x8); }
        }
        for (int i = 0; i < 4; i++)
            switch (L->sub_part[i]) {
            case PCAMV_D_L0_8x8: cb_mvd(L, C, 4 * i, 2, 2); break;
            case PCAMV_D_L0_8x4: cb_mvd(L, C, 4 * i, 2, 1); cb_mvd(L, C, 4 * i + 2, 2, 1); break;
            case PCAMV_D_L0_4x8: cb_mvd(L, C, 4 * i, 1, 2); cb_mvd(L, C, 4 * i + 1, 1, 2); break;
            default: for (int k = 0; k < 4; k++) cb_mvd(L, C, 4 * i + k, 1, 1); break;
            }
    } else if (L->i_partition == PCAMV_D_16x16) cb_mvd(L, C, 0, 4, 4);
    else if (L->i_partition == PCAMV_D_16x8) { cb_mvd(L, C, 0, 4, 2); cb_mvd(L, C, 8, 4, 2); }
    else { cb_mvd(L, C, 0, 2, 4); cb_mvd(L, C, 4, 2, 4); }
    const int cbp = L->cbp_luma, cl = L->cbp_left, ct = L->cbp_top;
    prim_cb_dec(L, C, 76 - ((cl >> 1) & 1) - ((ct >> 1) & 2), cbp & 1);
    prim_cb_dec(L, C, 76 - (cbp & 1) - ((ct >> 2) & 2), (cbp >> 1) & 1);
    prim_cb_dec(L, C, 76 - ((cl >> 3) & 1) - ((cbp << 1) & 2), (cbp >> 2) & 1);
    prim_cb_dec(L, C, 76 - ((cbp >> 2) & 1) - (cbp & 2), (cbp >> 3) & 1);
    const int ca = cl & 0x30, cb = ct & 0x30;
    prim_cb_dec(L, C, 77 + ((ca && cl != -1) ? 1 : 0) + ((cb && ct != -1) ? 2 : 0), L->cbp_chroma != 0);
    if (L->cbp_chroma) prim_cb_dec(L, C, 77 + 4 + (ca == 0x20) + 2 * (cb == 0x20), L->cbp_chroma > 1);
    if (L->cbp_luma | L->cbp_chroma) prim_cb_dec(L, C, 60, 0);
    (void)F;
}
/* x264_rd_cost_mb (rdo.c:139-171) of the macroblock as the cache describes it: distortion (SSD + psy-RD) + lambda2 * bits */
PCAMV_DEV int rd_cost_mb(const FrameDev &F, MBLocal *L)
{
    L->b_skip_mc = 0;
    const unsigned long long t_e = PROF_T();
#if defined(PCAMV_EXP_DBL) && PCAMV_EXP_DBL == 1      /* instruction-count experiments only: one part of a trial made twice */
    mb_encode(F, L, 0, 1);
#endif
    mb_encode(F, L, 0, 1);
    PROF_ADD(18, t_e);
    const unsigned long long t_s = PROF_T();
#if defined(PCAMV_EXP_DBL) && PCAMV_EXP_DBL == 2
    { volatile int sink = prim_ssd_mb(F, L); (void)sink; }
#endif
    const int ssd = prim_ssd_mb(F, L);
    PROF_ADD(19, t_s);
    int bits;
    if (F.b_cabac) {
        const unsigned long long t_h = PROF_T();
        CabWalk C;
#if defined(PCAMV_EXP_DBL) && PCAMV_EXP_DBL == 3
        { CabWalk C2; prim_cab_begin(L, C2, 1); cabac_mb_header(F, L, C2); volatile int sink = prim_cab_end(L, C2, 0); (void)sink; }
#endif
        prim_cab_begin(L, C, 1);                             /* a size trial: the slice's states are read, never written */
        cabac_mb_header(F, L, C);
        PROF_ADD(20, t_h);
        const unsigned long long t_r = PROF_T();
#if defined(PCAMV_EXP_DBL) && PCAMV_EXP_DBL == 4
        { CabWalk C2 = C; prim_cab_residual(F, L, C2, 0); volatile int sink = C2.vbits; (void)sink; }
#endif
        prim_cab_residual(F, L, C, 0);
        const int f8 = prim_cab_end(L, C, 0);
        PROF_ADD(21, t_r);
        bits = (int)(((unsigned long long)(unsigned)f8 * (unsigned long long)F.lambda2 + 32768ull) >> 16);
    } else
        bits = (int)((unsigned)prim_cavlc_mb(F, L) * (unsigned)F.lambda2 + 128u) >> 8;       /* int arithmetic in the reference */
    return ssd + bits;
}
/* x264_rd_cost_part for one 8x8 of a P_8x8 macroblock (rdo.c:202-245, i_pixel = PIXEL_8x8): x264_macroblock_encode_p8x8 of that
 * 8x8 with the sub-partition the cache describes (encoder/macroblock.c:929-1052), SSD + psy of its luma 8x8 and plain SSD of its two
 * chroma 4x4, x264_partition_size_cabac / _cavlc (encoder/cabac.c:1032-1075, cavlc.c:621-661: the sub-partition's MV differences,
 * the luma blocks when coded, always the two chroma AC blocks); (ssd << 8) + bits at 8 more bits of precision */
PCAMV_DEV unsigned long long rd_cost_part8(const FrameDev &F, MBLocal *L, int i8)
{
    L->b_skip_mc = 0;
    L->cbp_luma = 0;
    prim_encode_p8x8(F, L, i8);
    const unsigned long long ssd = (unsigned long long)(unsigned)prim_ssd_part8(F, L, i8);
    unsigned long long bits;
    if (F.b_cabac) {
        CabWalk C;
        prim_cab_begin(L, C, 1);
        switch (L->sub_part[i8]) {           /* x264_cabac_mb8x8_mvd */
        case PCAMV_D_L0_8x8: cb_mvd(L, C, 4 * i8, 2, 2); break;
        case PCAMV_D_L0_8x4: cb_mvd(L, C, 4 * i8, 2, 1); cb_mvd(L, C, 4 * i8 + 2, 2, 1); break;
        case PCAMV_D_L0_4x8: cb_mvd(L, C, 4 * i8, 1, 2); cb_mvd(L, C, 4 * i8 + 1, 1, 2); break;
        default: for (int k = 0; k < 4; k++) cb_mvd(L, C, 4 * i8 + k, 1, 1); break;
        }
        prim_cab_residual_part(F, L, C, i8);
        const int f8 = prim_cab_end(L, C, 0);
        bits = ((unsigned long long)(unsigned)f8 * (unsigned long long)F.lambda2 + 128ull) >> 8;
    } else
        bits = (unsigned long long)(unsigned)prim_cavlc_part8(F, L, i8) * (unsigned long long)F.lambda2;
#if defined(PCAMV_TRACE) && !defined(PCAMV_HOST_EMU)      /* diagnostics build only */
    if (F.trace && L->mb_xy == F.trace_mb && LANE() == 0) {
        int k = F.trace[0];
        if (k < 4000) { int *t = F.trace + 1 + 8 * k; t[0] = -1; t[1] = i8; t[2] = L->sub_part[i8]; t[3] = (int)ssd; t[4] = (int)bits; t[5] = L->cbp_luma; t[6] = L->cmv[scan8_of(4 * i8)][0]; t[7] = L->cmv[scan8_of(4 * i8)][1]; F.trace[0] = k + 1; }
    }
#endif
    return (ssd << 8) + bits;
}
/* One RD trial of the mode the cache describes.  The trials run in the order the decision compares them (16x16, 16x8, 8x16, 8x8,
 * strict <), so the one that is the cheapest so far is the decision's winner as far as it has come: its products are kept
 * (prim_rd_keep), and the macroblock as decided is not encoded and walked a second time when it is the kept one (mbk_search).
 * counts = 0: a trial the decision will not look at (P_8x8 while nothing is embedded, analyse.c:2841). */
PCAMV_DEV int rd_trial(const FrameDev &F, MBLocal *L, int counts)
{
#ifdef PCAMV_EXP_DOUBLE_TRIAL      /* instruction-count experiment only: every RD trial made twice (same result) */
    { volatile int sink = rd_cost_mb(F, L); (void)sink; }
#endif
    const int cost = rd_cost_mb(F, L);
    if (counts && cost < L->snap_cost) { L->snap_cost = cost; L->snap_part = L->i_partition; prim_rd_keep(F, L); }
    return cost;
}
template <int TESA>
PCAMV_DEV void analyse_p_rd(const FrameDev &F, MBLocal *L, struct Analysis *a, int i_satd)
{
    const int thresh = i_satd * 5 / 4;
    L->i_type = PCAMV_P_L0;
    if (a->rd16x16 == PCAMV_COST_MAX && a->me16x16.cost <= i_satd * 3 / 2) { L->i_partition = PCAMV_D_16x16; update_cache(L, a); a->rd16x16 = rd_trial(F, L, 1); }
    a->me16x16.cost = a->rd16x16;
    if (a->cost16x8 <= thresh) { L->i_partition = PCAMV_D_16x8; update_cache(L, a); a->cost16x8 = rd_trial(F, L, 1); } else a->cost16x8 = PCAMV_COST_MAX;
    if (a->cost8x16 <= thresh) { L->i_partition = PCAMV_D_8x16; update_cache(L, a); a->cost8x16 = rd_trial(F, L, 1); } else a->cost8x16 = PCAMV_COST_MAX;
    if (a->cost8x8 <= thresh) {
        L->i_type = PCAMV_P_8x8; L->i_partition = PCAMV_D_8x8;
        /* (the flag is read again here, through an empty asm: in the --me tesa instance hipcc (ROCm 7.2) tested a copy of "no sub-8x8
         * partitions" it had made ~60 k instructions earlier in a caller-saved scalar pair, s[40:41], which the search function
         * called in between uses as scratch -- the branch went the wrong way with the flag itself intact) */
        unsigned inter_now = F.inter;
#ifndef PCAMV_HOST_EMU
        asm volatile("" : "+s"(inter_now));
#endif
        if ((TESA & 8) && (inter_now & PCAMV_ANALYSE_PSUB8x8)) {
            /* analyse.c:2150-2180: per 8x8 the sub-partition shapes whose SATD cost is within 5/4 of the best are priced with
             * x264_rd_cost_part, the 8x8 shape itself only if another one was.  No update of the whole cache here: it holds what the
             * last trial / search left, and the trials of one 8x8 see what the others' left behind (non-zero flags, MV differences) */
            for (int i = 0; i < 4; i++) {
                const int c0 = a->cost4x4[i], c1 = a->cost8x4[i], c2 = a->cost4x8[i], c3 = a->me8x8[i].cost;
                const int th = imin(imin(c0, c1), imin(c2, c3)) * 5 / 4;
                int btype = PCAMV_D_L0_8x8;
                unsigned long long bcost = ~0ull;
                for (int subtype = PCAMV_D_L0_4x4; subtype <= PCAMV_D_L0_8x8; subtype++) {
                    const int c = subtype == 0 ? c0 : subtype == 1 ? c1 : subtype == 2 ? c2 : c3;
                    if (c > th || (subtype == PCAMV_D_L0_8x8 && bcost == ~0ull)) continue;
                    L->sub_part[i] = (uint8_t)subtype;
                    cache_mv_p8x8(L, a, i);
                    const unsigned long long cost = rd_cost_part8(F, L, i);
                    if (cost < bcost) { bcost = cost; btype = subtype; }
                }
                L->sub_part[i] = (uint8_t)btype;
                cache_mv_p8x8(L, a, i);
            }
        } else update_cache(L, a);
        a->cost8x8 = rd_trial(F, L, F.embed);
    } else a->cost8x8 = PCAMV_COST_MAX;
}
/* what the entropy coder leaves behind for the following macroblocks (encoder.c:1900-1927, common/macroblock.c:1254-1400):
 * the context states adapted to the macroblock as coded, its non-zero flags / counts, coded block pattern and MV differences */
/* walked = 1: the macroblock layer was walked already by the kept trial, whose end states, flags and MV differences have been
 * restored (prim_rd_restore); what is left is the mb_skip_flag decision, whose context nothing else in a macroblock touches */
PCAMV_DEV void entropy_commit(const FrameDev &F, MBLocal *L, int walked)
{
    const int skip = L->i_type == PCAMV_P_SKIP;
    if (F.b_cabac) {
        CabWalk C;
        prim_cab_begin(L, C, 0);
        prim_cb_dec(L, C, 11 + (L->type_left >= 0 && L->type_left != PCAMV_P_SKIP) + (L->type_top >= 0 && L->type_top != PCAMV_P_SKIP), skip);   /* x264_cabac_mb_skip */
        if (!skip && !walked) { cabac_mb_header(F, L, C); prim_cab_residual(F, L, C, 1); }
        prim_cab_end(L, C, 1);                               /* the macroblock as coded: the adapted states are the slice's */
    } else if (!skip && !walked) prim_cavlc_mb(F, L);
    prim_rd_commit(F, L, skip);
}

template <int TESA>
PCAMV_DEV int analyse_p16x16(const FrameDev &F, MBLocal *L, Analysis *a, int b_try_pskip)
{
    MEState me;
    me_setup(&me, PIX_16x16, 0, 0);
    predict_mv_16x16(L, 0, me.mvp);
    for (int i = 0; i < 9; i++) { L->mvc16[i][0] = 0; L->mvc16[i][1] = 0; }
    int i_mvc = predict_mv_ref16x16(F, L, L->mvc16);
    me_search<TESA>(F, L, &me, L->mvc16, i_mvc);
    if (b_try_pskip && me.cost - me.cost_mv < 300 * F.lambda &&
        iabs(me.mv[0] - L->pskip_mv[0]) + iabs(me.mv[1] - L->pskip_mv[1]) <= 1 && probe_pskip(F, L)) {
        L->i_type = PCAMV_P_SKIP;
        update_cache(L, a);
        return 1;
    }
    a->me16x16 = me;
    a->mvc[0][0] = me.mv[0]; a->mvc[0][1] = me.mv[1];
    prim_store_mvr(F, L, me.mv[0], me.mv[1]);
    L->mvr_own[0] = (int16_t)me.mv[0]; L->mvr_own[1] = (int16_t)me.mv[1];
    cache_ref_set(L, 0, 0, 4, 4, 0);
    L->i_type = PCAMV_P_L0;
    if (MBRD_ON) {                                   /* analyse.c:1194-1203 */
        prim_fenc_complexity(F, L);
        /* the reference makes this trial here; nothing the searches that follow read depends on it (it only sets i_rd16x16), so it
         * is made at the head of the RD stage (analyse_decide), which keeps everything that needs the entropy coder's state of the
         * macroblock coded before this one -- the chain from macroblock to macroblock -- in one place behind the searches */
        a->rd16_early = me.mv[0] == L->pskip_mv[0] && me.mv[1] == L->pskip_mv[1];
    }
    return 0;
}
template <int TESA>
PCAMV_DEV void analyse_p8x8(const FrameDev &F, MBLocal *L, Analysis *a)
{
    int i_mvc = 1;
    L->i_partition = PCAMV_D_8x8;
    a->mvc[0][0] = a->me16x16.mv[0]; a->mvc[0][1] = a->me16x16.mv[1];
    for (int i = 0; i < 4; i++) {
        MEState *me = &a->me8x8[i];
        int x8 = i % 2, y8 = i / 2;
        me_setup(me, PIX_8x8, 8 * x8, 8 * y8);
        predict_mv(L, 4 * i, 2, me->mvp);
        me_search<TESA>(F, L, me, a->mvc, i_mvc);
        cache_mv_set(L, 2 * x8, 2 * y8, 2, 2, me->mv[0], me->mv[1]);
        a->mvc[i_mvc][0] = me->mv[0]; a->mvc[i_mvc][1] = me->mv[1]; i_mvc++;
        me->cost += F.lambda * 1;
    }
    a->cost8x8 = a->me8x8[0].cost + a->me8x8[1].cost + a->me8x8[2].cost + a->me8x8[3].cost;
    for (int i = 0; i < 4; i++) L->sub_part[i] = PCAMV_D_L0_8x8;
}
template <int TESA>
PCAMV_DEV void analyse_p16x8(const FrameDev &F, MBLocal *L, Analysis *a)
{
    L->i_partition = PCAMV_D_16x8;
    for (int i = 0; i < 2; i++) {
        MEState me; int mvc[3][2];
        me_setup(&me, PIX_16x8, 0, 8 * i);
        mvc[0][0] = a->mvc[0][0]; mvc[0][1] = a->mvc[0][1];
        mvc[1][0] = a->mvc[2 * i + 1][0]; mvc[1][1] = a->mvc[2 * i + 1][1];
        mvc[2][0] = a->mvc[2 * i + 2][0]; mvc[2][1] = a->mvc[2 * i + 2][1];
        cache_ref_set(L, 0, 2 * i, 4, 2, 0);
        predict_mv(L, 8 * i, 4, me.mvp);
        me_search<TESA>(F, L, &me, mvc, 3);
        a->me16x8[i] = me;
        cache_mv_set(L, 0, 2 * i, 4, 2, me.mv[0], me.mv[1]);
    }
    a->cost16x8 = a->me16x8[0].cost + a->me16x8[1].cost;
}
template <int TESA>
PCAMV_DEV void analyse_p8x16(const FrameDev &F, MBLocal *L, Analysis *a)
{
    L->i_partition = PCAMV_D_8x16;
    for (int i = 0; i < 2; i++) {
        MEState me; int mvc[3][2];
        me_setup(&me, PIX_8x16, 8 * i, 0);
        mvc[0][0] = a->mvc[0][0]; mvc[0][1] = a->mvc[0][1];
        mvc[1][0] = a->mvc[i + 1][0]; mvc[1][1] = a->mvc[i + 1][1];
        mvc[2][0] = a->mvc[i + 3][0]; mvc[2][1] = a->mvc[i + 3][1];
        cache_ref_set(L, 2 * i, 0, 2, 4, 0);
        predict_mv(L, 4 * i, 2, me.mvp);
        me_search<TESA>(F, L, &me, mvc, 3);
        a->me8x16[i] = me;
        cache_mv_set(L, 2 * i, 0, 2, 4, me.mv[0], me.mv[1]);
    }
    a->cost8x16 = a->me8x16[0].cost + a->me8x16[1].cost;
}
template <int TESA>
PCAMV_DEV void analyse_sub8x8(const FrameDev &F, MBLocal *L, Analysis *a, int i8, int pixel)
{
    L->i_partition = PCAMV_D_8x8;
    int n = pixel == PIX_4x4 ? 4 : 2, cost = 0;
    for (int k = 0; k < n; k++) {
        int idx = 4 * i8 + (pixel == PIX_8x4 ? 2 * k : k);
        MEState *me = pixel == PIX_4x4 ? &a->me4x4[i8][k] : pixel == PIX_8x4 ? &a->me8x4[i8][k] : &a->me4x8[i8][k];
        me_setup(me, pixel, 4 * blk_x_of(idx), 4 * blk_y_of(idx));
        predict_mv(L, idx, pixel == PIX_8x4 ? 2 : 1, me->mvp);
        int mvc[1][2];
        const MEState *cand = pixel == PIX_4x4 ? &a->me8x8[i8] : &a->me4x4[i8][0];
        mvc[0][0] = cand->mv[0]; mvc[0][1] = cand->mv[1];
        me_search<TESA>(F, L, me, mvc, k == 0);
        cache_mv_set(L, blk_x_of(idx), blk_y_of(idx), pixel == PIX_8x4 ? 2 : 1, pixel == PIX_4x8 ? 2 : 1, me->mv[0], me->mv[1]);
        cost += me->cost;
    }
    cost += F.lambda * (pixel == PIX_4x4 ? 5 : 3);
    if (F.b_chroma_me) {
        /* analyse.c:1535-1567: chroma of the 8x8 predicted with one MV per luma 4x4 */
        int qx[4], qy[4];
        for (int q = 0; q < 4; q++) {
            const MEState *m4 = pixel == PIX_4x4 ? &a->me4x4[i8][q] : pixel == PIX_8x4 ? &a->me8x4[i8][q >> 1] : &a->me4x8[i8][q & 1];
            qx[q] = m4->mv[0]; qy[q] = m4->mv[1];
        }
        cost += prim_chroma4x4_cost(F, L, i8, qx, qy, F.subme > 1);
    }
    if (pixel == PIX_4x4) a->cost4x4[i8] = cost; else if (pixel == PIX_8x4) a->cost8x4[i8] = cost; else a->cost4x8[i8] = cost;
}

/* d_mv: {0,-1},{1,0},{0,1},{-1,0},{-2,1},{-1,2},{1,2},{2,1},{2,-1},{1,-2},{-1,-2},{-2,-1}   d_nb: {0,-1},{1,0},{0,1},{-1,0},{-1,-1},{-1,1},{1,-1},{1,1},{0,0} */
PCAMV_DEV int d_mv_x(int i) { return nib64(0x013443101232ull, i) - 2; }
PCAMV_DEV int d_mv_y(int i) { return nib64(0x100134432321ull, i) - 2; }
PCAMV_DEV int d_nb_x(int i) { return nib64(0x122000121ull, i) - 1; }
PCAMV_DEV int d_nb_y(int i) { return nib64(0x120201210ull, i) - 1; }

/* nine neighbourhood costs around (cx,cy) on the current reconstruction (MV_SATD_FDEC_IH: metric of
 * the reconstructed block against the reference at the candidate MV + MV bits + chroma for
 * partitions >= 8x8), one list; returns the minimum, the centre's cost in *last */
PCAMV_DEV int rca_nine(const FrameDev &F, MBLocal *L, MEState *me, const uint8_t *enc, int cx, int cy, int nb_cost, int *last, int win)
{
    const int flags = (F.subme > 1 ? EV_SATD : 0) | ((F.b_chroma_me && me->i_pixel <= PIX_8x8) ? EV_CHROMA : 0) | (win ? EV_WIN : 0);
    FOR_CAND(c, 9) L->cxy[c] = CAND_PACK(cx + d_nb_x(c), cy + d_nb_y(c));
    EvalRes r = eval_cands(F, L, me, enc, 9, flags);
    if (nb_cost) { FOR_CAND(c, 9) L->nbc[c] = L->ccost[c]; PCAMV_WAVE_SYNC(); }
    *last = L->ccost[8];
    return r.cost;
}

/* have_base: L->recb0 already holds the reconstruction of the macroblock as decided (the first re-encode
 * of x264_ih_get_mv_cost is the same for every carrier of the macroblock, and is its pass-1 output) */
PCAMV_DEV int rca_mv_cost(const FrameDev &F, MBLocal *L, Analysis *a, MEState *me, int *m_x, int *m_y, int have_base, int win = 0)
{
    const float beta1 = 1.4, beta2 = 4;
    int bmx = me->mv[0], bmy = me->mv[1];
    int cost = 0, min_cost;
    int b_1_neighbor = 0, b_error_pos = 0;
    if (!have_base) { update_cache(L, a); mb_encode(F, L); prim_copy_pred(L, L->recb0); }
    const unsigned long long t_n = PROF_T();
    min_cost = rca_nine(F, L, me, L->recb0, bmx, bmy, 1, &cost, win);
    PROF_ADD(35, t_n);
    me->cost_rec = L->nbc[8];
    const int want_optimal = !(min_cost < me->cost_rec);
    min_cost = PCAMV_COST_MAX; *m_x = 0; *m_y = 0;
    int ii_best = -1;
    if (win) {
        /* 16x16 macroblock with its reference window in LDS: the four replacement MVs of a group are
         * re-encoded together (their evaluations do not depend on each other, only the fold below does)
         * and their 4 x 9 neighbourhood costs are one list, candidate c = 4 k + j for re-encode j,
         * neighbour k; groups: ii = 0..3 always, 4..7 and 8..11 only when none of 0..3 qualified */
        const int flags = (F.subme > 1 ? EV_SATD : 0) | (F.b_chroma_me ? EV_CHROMA : 0) | EV_WIN | EV_SRC4;
        for (int grp = 0; grp < 3; grp++) {
            PROF_CNT(40, 1);
            const unsigned long long t_g = PROF_T();
            for (int j = 0; j < 4; j++) prim_predict_win16(F, L, j, bmx + d_mv_x(4 * grp + j), bmy + d_mv_y(4 * grp + j));
            prim_mb_transform4(F, L);
            PROF_ADD(36, t_g);
            const unsigned long long t_l = PROF_T();
            FOR_CAND(c, 36) {
                int k = c >> 2, ii = 4 * grp + (c & 3);
                L->cxy[c] = CAND_PACK(bmx + d_mv_x(ii) + d_nb_x(k), bmy + d_mv_y(ii) + d_nb_y(k));
            }
            eval_cands(F, L, me, L->pred4[0], 36, flags);
            PROF_ADD(37, t_l);
            for (int j = 0; j < 4; j++) {
                const int ii = 4 * grp + j;
                int min1 = PCAMV_COST_MAX;
                for (int k = 0; k < 9; k++) min1 = imin(min1, L->ccost[4 * k + j]);
                cost = L->ccost[4 * 8 + j];
                int is_opt = (min1 == cost);
                if (is_opt == want_optimal && cost < min_cost) { min_cost = cost; *m_x = d_mv_x(ii); *m_y = d_mv_y(ii); ii_best = ii; }
            }
            if (grp == 0 && min_cost != PCAMV_COST_MAX) break;
        }
    } else
    for (int ii = 0; ii < 12; ii++) {
        PROF_CNT(41, 1);
        int bx1 = bmx + d_mv_x(ii), by1 = bmy + d_mv_y(ii);
        me->mv[0] = bx1; me->mv[1] = by1;
        update_cache(L, a); mb_encode(F, L, win); prim_copy_pred(L, L->recb);
        int min1 = rca_nine(F, L, me, L->recb, bx1, by1, 0, &cost, win);
        int is_opt = (min1 == cost);
        if (is_opt == want_optimal && cost < min_cost) { min_cost = cost; *m_x = d_mv_x(ii); *m_y = d_mv_y(ii); ii_best = ii; }
        if (ii == 3 && min_cost != PCAMV_COST_MAX) break;
    }
    if (min_cost == PCAMV_COST_MAX) {
        b_error_pos = 1; b_1_neighbor = 1;
        *m_x = 0; *m_y = 0;
        for (int k = 0; k < 4; k++) if (L->nbc[k] < min_cost) { min_cost = L->nbc[k]; *m_x = d_nb_x(k); *m_y = d_nb_y(k); }
    } else b_1_neighbor = ii_best <= 3;
    int cost_opt = min_cost > me->cost_rec ? min_cost - me->cost_rec : 1;
    if (!b_1_neighbor) cost_opt = (int)(beta1 * (float)cost_opt);
    else if (b_error_pos) cost_opt = (int)(beta2 * (float)cost_opt);
    me->mv[0] = bmx; me->mv[1] = bmy;
    update_cache(L, a);
    return cost_opt;
}

/* carrier slots of a record in embedding order (encoder.c:1566-1647) */
PCAMV_DEV int carrier_slots(int i_type, int i_partition, const uint8_t *sub, int used, int slots[16])
{
    int n = 0;
    if (!used) return 0;
    if (i_type == PCAMV_P_8x8) {
        for (int i = 0; i < 4; i++)
            switch (sub[i]) {
            case PCAMV_D_L0_8x8: slots[n++] = i * 4; break;
            case PCAMV_D_L0_4x8: slots[n++] = i * 4; slots[n++] = i * 4 + 1; break;
            case PCAMV_D_L0_8x4: slots[n++] = i * 4; slots[n++] = i * 4 + 2; break;
            default: for (int j = 0; j < 4; j++) slots[n++] = i * 4 + j; break;
            }
    } else if (i_type == PCAMV_P_L0) {
        if (i_partition == PCAMV_D_16x16) slots[n++] = 0;
        else if (i_partition == PCAMV_D_8x16) { slots[n++] = 0; slots[n++] = 4; }
        else if (i_partition == PCAMV_D_16x8) { slots[n++] = 0; slots[n++] = 8; }
    }
    return n;
}

/* the carrier slot that owns 4x4 block i (x264 block order) */
PCAMV_DEV int carrier_of_block(int i_type, int i_partition, const uint8_t *sub, int i)
{
    if (i_type == PCAMV_P_8x8) {
        const int i8 = i >> 2, j = i & 3;
        switch (sub[i8]) {
        case PCAMV_D_L0_8x8: return 4 * i8;
        case PCAMV_D_L0_4x8: return 4 * i8 + (j & 1);
        case PCAMV_D_L0_8x4: return 4 * i8 + (j & 2);
        default: return i;
        }
    }
    if (i_partition == PCAMV_D_8x16) return blk_x_of(i) < 2 ? 0 : 4;
    if (i_partition == PCAMV_D_16x8) return blk_y_of(i) < 2 ? 0 : 8;
    return 0;
}

/* ---------------------------------------------------------------- phase A: search + decision */
/* Analyse one macroblock (motion search, partition decision, early skip) and publish its final
 * motion to the frame arrays its right/lower neighbours read.  Writes the record without the
 * RCA fields; the search-time mvp of every carrier slot goes to mvp_aux for phase B.
 * Three stages, so that a schedule can hand the macroblock's successor its motion as early as it is known (the speculative
 * raster schedule of the RD instance, pcamv_kernels.hip.h):
 *   analyse_s16     early P_SKIP, else the 16x16 search (returns 1 for a skipped macroblock: nothing else follows)
 *   analyse_s_rest  the other partitions' searches and the decision by SATD cost
 *   analyse_decide  quarter-pel refinement of the decided partition, or -- --subme >= 6 -- the intra thresholds and the RD trials:
 *                   the only stage that reads what the entropy coder left behind in the macroblock coded before this one */
template <int TESA>
PCAMV_DEV int analyse_s16(const FrameDev &F, MBLocal *L, Analysis *a)
{
    int b_skip = 0, b_try_pskip = 0;
    for (int i = 0; i < 4; i++) L->sub_part[i] = PCAMV_D_L0_8x8;
    L->i_partition = PCAMV_D_16x16;
    a->rd16x16 = a->cost8x8 = a->cost16x8 = a->cost8x16 = PCAMV_COST_MAX;      /* analyse.c:321-332 */
    for (int i = 0; i < 4; i++) a->cost4x4[i] = a->cost8x4[i] = a->cost4x8[i] = PCAMV_COST_MAX;      /* (read by the RD stage whether analysed or not) */
    a->rd16_early = 0;
    L->snap_part = -1; L->snap_cost = PCAMV_COST_MAX;
    if (F.b_fast_pskip) {
        if (F.subme >= 3) b_try_pskip = 1;
        else if (L->type_left == PCAMV_P_SKIP || L->type_top == PCAMV_P_SKIP || L->type_topleft == PCAMV_P_SKIP || L->type_topright == PCAMV_P_SKIP)
            b_skip = probe_pskip(F, L);
    }
    if (b_skip) { L->i_type = PCAMV_P_SKIP; L->i_partition = PCAMV_D_16x16; return 1; }
    return analyse_p16x16<TESA>(F, L, a, b_try_pskip);
}
template <int TESA>
PCAMV_DEV void analyse_s_rest(const FrameDev &F, MBLocal *L, Analysis *a)
{
    int i_cost;
    const unsigned flags = F.inter;
    int i_type = PCAMV_P_L0, i_partition = PCAMV_D_16x16;
    const unsigned long long t_b = PROF_T();
    if (flags & PCAMV_ANALYSE_PSUB16x16) analyse_p8x8<TESA>(F, L, a);
    PROF_ADD(7, t_b);
    const unsigned long long t_c = PROF_T();
    i_cost = a->me16x16.cost;
    if ((flags & PCAMV_ANALYSE_PSUB16x16) && a->cost8x8 < a->me16x16.cost) {
        if (flags & PCAMV_ANALYSE_PSUB8x8) {
            i_type = PCAMV_P_8x8; i_partition = PCAMV_D_8x8; i_cost = a->cost8x8;
            for (int i = 0; i < 4; i++) {
                analyse_sub8x8<TESA>(F, L, a, i, PIX_4x4);
                if (a->cost4x4[i] < a->me8x8[i].cost) {
                    int c8 = a->cost4x4[i];
                    L->sub_part[i] = PCAMV_D_L0_4x4;
                    analyse_sub8x8<TESA>(F, L, a, i, PIX_8x4);
                    if (a->cost8x4[i] < c8) { c8 = a->cost8x4[i]; L->sub_part[i] = PCAMV_D_L0_8x4; }
                    analyse_sub8x8<TESA>(F, L, a, i, PIX_4x8);
                    if (a->cost4x8[i] < c8) { c8 = a->cost4x8[i]; L->sub_part[i] = PCAMV_D_L0_4x8; }
                    i_cost += c8 - a->me8x8[i].cost;
                }
                cache_mv_p8x8(L, a, i);
            }
            a->cost8x8 = i_cost;
        }
    }
    if ((flags & PCAMV_ANALYSE_PSUB16x16) && a->cost8x8 < a->me16x16.cost + a->me8x8[1].cost_mv + a->me8x8[2].cost_mv) {
        analyse_p16x8<TESA>(F, L, a);
        if (a->cost16x8 < i_cost) { i_cost = a->cost16x8; i_type = PCAMV_P_L0; i_partition = PCAMV_D_16x8; }
        analyse_p8x16<TESA>(F, L, a);
        if (a->cost8x16 < i_cost) { i_cost = a->cost8x16; i_type = PCAMV_P_L0; i_partition = PCAMV_D_8x16; }
    }
    L->i_partition = i_partition;
    a->sel_type = i_type; a->sel_part = i_partition; a->sel_cost = i_cost;
    PROF_ADD(8, t_c);
}
template <int TESA>
PCAMV_DEV void analyse_decide(const FrameDev &F, MBLocal *L, Analysis *a)
{
    int i_type = a->sel_type, i_partition = a->sel_part, i_cost = a->sel_cost;
    const unsigned long long t_d = PROF_T();
    if (MBRD_ON) {
        /* analyse.c:2749-2752, 2809-2850: no quarter-pel refinement; the intra SATD cost (never an intra mode: analyse.c:2863)
         * bounds the RD trials; x264_rd_cost_mb decides the partition; P_8x8 only while embedding (analyse.c:2841) */
        int i16, i4;
        if (a->rd16_early) {                 /* analyse.c:1197-1202, see analyse_p16x16 */
            /* (the reference's cache had these MVs BEFORE the other searches overwrote them; here the searches' MVs are put back
             * after the trial: the sub-partition RD trials predict MVs from whatever the cache holds, analyse.c:2150) */
            uint32_t keep_mv[16];
            PCAMV_WAVE_SYNC();
            FOR_CAND(i, 16) keep_mv[NB_SLOT(i)] = ((const uint32_t *)L->cmv)[SCAN8_0 + (i & 3) + 8 * (i >> 2)];
            const int part_keep = L->i_partition;
            L->i_type = PCAMV_P_L0; L->i_partition = PCAMV_D_16x16;
            cache_mv_set(L, 0, 0, 4, 4, a->me16x16.mv[0], a->me16x16.mv[1]);
            a->rd16x16 = rd_trial(F, L, 1);
            PCAMV_WAVE_SYNC();
            FOR_CAND(i, 16) ((uint32_t *)L->cmv)[SCAN8_0 + (i & 3) + 8 * (i >> 2)] = keep_mv[NB_SLOT(i)];
            PCAMV_WAVE_SYNC();
            L->i_partition = part_keep;
        }
        const unsigned long long t_i = PROF_T();
        if (F.b_chroma_me) {
            const int c8 = intra_chroma_cost(F, L);
            intra_analyse(F, L, i_cost - c8, &i16, &i4);
            i16 += c8; i4 += c8;
        } else intra_analyse(F, L, i_cost, &i16, &i4);
        PROF_ADD(16, t_i);
        const unsigned long long t_rd = PROF_T();
        analyse_p_rd<TESA>(F, L, a, imin(i_cost, imin(i16, i4)));
        PROF_ADD(17, t_rd);
        i_type = PCAMV_P_L0; i_partition = PCAMV_D_16x16; i_cost = a->me16x16.cost;
        if (a->cost16x8 < i_cost) { i_cost = a->cost16x8; i_partition = PCAMV_D_16x8; }
        if (a->cost8x16 < i_cost) { i_cost = a->cost8x16; i_partition = PCAMV_D_8x16; }
        if (F.embed && a->cost8x8 < i_cost) { i_cost = a->cost8x8; i_partition = PCAMV_D_8x8; i_type = PCAMV_P_8x8; }
        L->i_partition = i_partition;
    } else
    if (i_partition == PCAMV_D_16x16) me_refine_qpel<TESA>(F, L, &a->me16x16);
    else if (i_partition == PCAMV_D_16x8) { me_refine_qpel<TESA>(F, L, &a->me16x8[0]); me_refine_qpel<TESA>(F, L, &a->me16x8[1]); }
    else if (i_partition == PCAMV_D_8x16) { me_refine_qpel<TESA>(F, L, &a->me8x16[0]); me_refine_qpel<TESA>(F, L, &a->me8x16[1]); }
    else
        for (int i = 0; i < 4; i++)
            switch (L->sub_part[i]) {
            case PCAMV_D_L0_8x8: me_refine_qpel<TESA>(F, L, &a->me8x8[i]); break;
            case PCAMV_D_L0_8x4: me_refine_qpel<TESA>(F, L, &a->me8x4[i][0]); me_refine_qpel<TESA>(F, L, &a->me8x4[i][1]); break;
            case PCAMV_D_L0_4x8: me_refine_qpel<TESA>(F, L, &a->me4x8[i][0]); me_refine_qpel<TESA>(F, L, &a->me4x8[i][1]); break;
            default: for (int k = 0; k < 4; k++) me_refine_qpel<TESA>(F, L, &a->me4x4[i][k]); break;
            }
    L->i_type = i_type;
    PROF_ADD(9, t_d);
}
template <int TESA>
PCAMV_DEV void analyse_mb_search(const FrameDev &F, MBLocal *L, Analysis *a)
{
    const unsigned long long t_a = PROF_T();
    const int skip = analyse_s16<TESA>(F, L, a);
    PROF_ADD(6, t_a);
    if (!skip) {
        analyse_s_rest<TESA>(F, L, a);
        analyse_decide<TESA>(F, L, a);
    }
    update_cache(L, a);
}

/* the MEState that owns carrier slot s of the decided partitioning */
PCAMV_DEV MEState *slot_me(MBLocal *L, Analysis *a, int slot)
{
    if (L->i_type == PCAMV_P_8x8) {
        int i = slot >> 2, j = slot & 3;
        switch (L->sub_part[i]) {
        case PCAMV_D_L0_8x8: return &a->me8x8[i];
        case PCAMV_D_L0_4x8: return &a->me4x8[i][j];
        case PCAMV_D_L0_8x4: return &a->me8x4[i][j >> 1];
        default: return &a->me4x4[i][j];
        }
    }
    if (L->i_partition == PCAMV_D_16x16) return &a->me16x16;
    if (L->i_partition == PCAMV_D_8x16) return &a->me8x16[slot >> 2];
    return &a->me16x8[slot >> 3];
}
/* geometry of carrier slot s: pixel type and offset inside the MB */
PCAMV_DEV void slot_geometry(int i_type, int i_partition, const uint8_t *sub, int slot, int *ip, int *xoff, int *yoff)
{
    if (i_type == PCAMV_P_8x8) {
        int i = slot >> 2;
        *xoff = 4 * blk_x_of(slot); *yoff = 4 * blk_y_of(slot);
        *ip = sub[i] == PCAMV_D_L0_8x8 ? PIX_8x8 : sub[i] == PCAMV_D_L0_4x8 ? PIX_4x8 : sub[i] == PCAMV_D_L0_8x4 ? PIX_8x4 : PIX_4x4;
    } else if (i_partition == PCAMV_D_16x16) { *ip = PIX_16x16; *xoff = 0; *yoff = 0; }
    else if (i_partition == PCAMV_D_8x16) { *ip = PIX_8x16; *xoff = slot ? 8 : 0; *yoff = 0; }
    else { *ip = PIX_16x8; *xoff = 0; *yoff = slot ? 8 : 0; }
}

#endif
