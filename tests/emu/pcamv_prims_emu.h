/*
 * pcamv_prims_emu.h -- TEST-ONLY scalar stand-ins for the lane-parallel GPU primitives, so that
 * the wave-uniform control code of the product (csrc/pcamv_logic.h) can be exercised on the CPU
 * (here, under sanitizers if wanted) against the oracle.  Never built into libpcamv_gpu.so.
 */
#ifndef PCAMV_PRIMS_EMU_H
#define PCAMV_PRIMS_EMU_H
#include <string.h>
#include <stddef.h>
#include "pcamv_common.h"

/* the reference's own tables (common/mc.c:194-200 hpel_ref0/1, dct.h zigzag, quant.c:203 decimate table) for the scalar restatement */
static const int hpel_ref0_tab[16] = {0, 1, 1, 1, 0, 1, 1, 1, 2, 3, 3, 3, 0, 1, 1, 1};
static const int hpel_ref1_tab[16] = {0, 0, 0, 0, 2, 2, 3, 2, 2, 2, 3, 2, 2, 2, 3, 2};
static const unsigned char zz4_tab[16] = {0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15};
static const unsigned char decimate_tab4[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

/* work counters for tools/dbg (lists, candidates and lane-passes by kind; macroblock re-encodes) */
static long long emu_stats[32];

static inline const uint8_t *emu_qpel(const FrameDev &F, uint8_t *tmp, int *st, int px, int py, int mvx, int mvy, int w, int h)
{
    int qidx = ((mvy & 3) << 2) + (mvx & 3);
    ptrdiff_t off = (ptrdiff_t)(py + (mvy >> 2)) * F.stride + px + (mvx >> 2);
    const uint8_t *a = F.luma[hpel_ref0_tab[qidx]] + off + ((mvy & 3) == 3) * F.stride;
    if (qidx & 5) {
        const uint8_t *b = F.luma[hpel_ref1_tab[qidx]] + off + ((mvx & 3) == 3);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) tmp[y * 32 + x] = (a[y * F.stride + x] + b[y * F.stride + x] + 1) >> 1;
        *st = 32; return tmp;
    }
    *st = F.stride; return a;
}
static inline int emu_had4(const uint8_t *a, int sa, const uint8_t *b, int sb)
{
    int d[4][4], t[4][4], s = 0;
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = a[y * sa + x] - b[y * sb + x];
    for (int y = 0; y < 4; y++) {
        int s01 = d[y][0] + d[y][1], d01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], d23 = d[y][2] - d[y][3];
        t[y][0] = s01 + s23; t[y][1] = d01 + d23; t[y][2] = s01 - s23; t[y][3] = d01 - d23;
    }
    for (int x = 0; x < 4; x++) {
        int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
        s += iabs(s01 + s23) + iabs(d01 + d23) + iabs(s01 - s23) + iabs(d01 - d23);
    }
    return s;
}
static inline int emu_cmp(int w, int h, const uint8_t *a, int sa, const uint8_t *b, int sb, int satd)
{
    int s = 0;
    if (!satd) { for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += iabs(a[y * sa + x] - b[y * sb + x]); return s; }
    if (w == 4) { for (int y = 0; y < h; y += 4) s += emu_had4(a + y * sa, sa, b + y * sb, sb) >> 1; return s; }
    for (int y = 0; y < h; y += 4) for (int x = 0; x < w; x += 8)
        s += (emu_had4(a + y * sa + x, sa, b + y * sb + x, sb) + emu_had4(a + y * sa + x + 4, sa, b + y * sb + x + 4, sb)) >> 1;
    return s;
}
static inline int prim_cost_luma_nolog(const FrameDev &F, MBLocal *L, const uint8_t *enc, int ip, int xoff, int yoff, int mx, int my, int satd)
{
    uint8_t tmp[32 * 20]; int st, w = pix_w_of(ip), h = pix_h_of(ip);
    const uint8_t *r = emu_qpel(F, tmp, &st, L->mb_x * 16 + xoff, L->mb_y * 16 + yoff, mx, my, w, h);
    return emu_cmp(w, h, enc + yoff * 16 + xoff, 16, r, st, satd);
}
static inline void emu_mc_chroma(const FrameDev &F, uint8_t *dst, int ds, int plane, int cx, int cy, int mvx, int mvy, int w, int h)
{
    int dx = mvx & 7, dy = mvy & 7, cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
    const uint8_t *s = F.chroma[plane] + (ptrdiff_t)(cy + (mvy >> 3)) * F.cstride + cx + (mvx >> 3);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
        dst[y * ds + x] = (cA * s[y * F.cstride + x] + cB * s[y * F.cstride + x + 1] + cC * s[(y + 1) * F.cstride + x] + cD * s[(y + 1) * F.cstride + x + 1] + 32) >> 6;
}
/* all listed candidates: pixel metric (+ MV bits) (+ chroma), costs to L->ccost, first minimum returned */
static inline EvalRes prim_eval_list(const FrameDev &F, MBLocal *L, const uint8_t *enc, int ip, int xoff, int yoff, int n, int flags, int mvp0, int mvp1)
{
    EvalRes r = {PCAMV_COST_MAX, -1};
    const int satd = flags & EV_SATD;
    { int kind = (flags & EV_FPEL) ? 0 : !satd ? 1 : !(flags & EV_CHROMA) ? 2 : 3, nblk = (pix_w_of(ip) >> 2) * (pix_h_of(ip) >> 2);
      emu_stats[kind]++; emu_stats[4 + kind] += n; emu_stats[8 + kind] += (n * nblk + 63) / 64;
      if (flags & EV_CHROMA) emu_stats[12] += (n * (nblk / 2) + 63) / 64; if (enc == L->recb || enc == L->recb0) emu_stats[13]++; }
    for (int c = 0; c < n; c++) {
        if (L->cxy[c] == CAND_NONE) { L->ccost[c] = PCAMV_COST_MAX; continue; }
        const int mx = CAND_X(c), my = CAND_Y(c);
        const uint8_t *src = (flags & EV_SRC4) ? enc + (c & 3) * 384 : enc;
        int cost = prim_cost_luma_nolog(F, L, src, ip, xoff, yoff, mx, my, satd);
        if (!(flags & EV_NOMV)) cost += F.cost_mv[mx - mvp0] + F.cost_mv[my - mvp1];
        if (flags & EV_CHROMA) {
            uint8_t tmp[8 * 8]; int w = pix_w_of(ip) / 2, h = pix_h_of(ip) / 2;
            for (int p = 0; p < 2; p++) {
                emu_mc_chroma(F, tmp, 8, p, L->mb_x * 8 + (xoff >> 1), L->mb_y * 8 + (yoff >> 1), mx, my, w, h);
                int cc = emu_cmp(w, h, src + 256 + (yoff >> 1) * 16 + p * 8 + (xoff >> 1), 16, tmp, 8, satd);
                if (flags & EV_PROBE) L->ccost[64 * (1 + p) + c] = cc; else cost += cc;
            }
        }
        L->ccost[c] = cost;
        if (cost < r.cost) { r.cost = cost; r.idx = c; }
        if (F.trace && L->mb_xy == F.trace_mb) { int k = F.trace[0]; if (k < 4000) { int *t = F.trace + 1 + 8 * k; t[0] = ip; t[1] = xoff; t[2] = yoff; t[3] = mx; t[4] = my; t[5] = flags | ((enc == L->recb || enc == L->recb0) ? 32 : 0); t[6] = cost; t[7] = c; F.trace[0] = k + 1; } }
    }
    return r;
}
/* exhaustive window: every full-pel position in raster order, first minimum */
static inline EvalRes prim_esa_window(const FrameDev &F, MBLocal *L, int ip, int xoff, int yoff, int min_x, int min_y, int width, int nrows, int mvp0, int mvp1)
{
    EvalRes r = {PCAMV_COST_MAX, -1};
    for (int ry = 0; ry < nrows; ry++)
        for (int rx = 0; rx < width; rx++) {
            const int mx = 4 * (min_x + rx), my = 4 * (min_y + ry);
            const int cost = prim_cost_luma_nolog(F, L, L->fenc, ip, xoff, yoff, mx, my, 0) + F.cost_mv[mx - mvp0] + F.cost_mv[my - mvp1];
            if (cost < r.cost) { r.cost = cost; r.idx = ry * width + rx; }
        }
    return r;
}
/* TESA row / walk / pruning (see pcamv_prims_gpu.h): the scalar statement, shaped like me.c:539-600 */
static inline void prim_tesa_row(const FrameDev &F, MBLocal *L, int ip, int xoff, int yoff, int min_x, int my, int width, int mvp0)
{
    const int bw = pix_w_of(ip), bh = pix_h_of(ip), sub = ip <= PIX_8x8 ? 8 : 4;
    for (int x = 0; x < width; x++) {
        const uint8_t *ref = F.luma[0] + (size_t)(L->mb_y * 16 + yoff + my) * F.stride + L->mb_x * 16 + xoff + min_x + x;
        int sad = 0, ads = 0;
        for (int r = 0; r < bh; r++) for (int c = 0; c < bw; c++) sad += iabs(L->fenc[(yoff + r) * 16 + xoff + c] - ref[(size_t)r * F.stride + c]);
        for (int sy = 0; sy < bh; sy += sub) for (int sx = 0; sx < bw; sx += sub) {
            int e = 0, rr = 0;
            for (int r = 0; r < sub; r++) for (int c = 0; c < sub; c++) { e += L->fenc[(yoff + sy + r) * 16 + xoff + sx + c]; rr += ref[(size_t)(sy + r) * F.stride + sx + c]; }
            ads += iabs(e - rr);
        }
        L->ccost[x] = sad + F.cost_mv[x * 4 - mvp0];                    /* me.c:551,563: index relative to the window (sic) */
        L->ccost[64 + x] = ads + F.cost_mv[(min_x + x) * 4 - mvp0];
    }
}
static inline int prim_tesa_scan(MBLocal *L, int width, int bsad, int sad_thresh, int ycost, int ry, int *n)
{
    const int thresh = bsad * 17 / 16;
    for (int x = 0; x < width; x++) {
        if (!(L->ccost[64 + x] < thresh)) continue;
        const int sad = L->ccost[x];
        if (sad < (bsad * sad_thresh >> 3)) {
            if (sad < bsad) bsad = sad;
            *TESA_SLOT(L, *n) = TESA_PACK(sad + ycost, ry, x); (*n)++;
        }
    }
    return bsad;
}
static inline int prim_tesa_select(MBLocal *L, int n, int limit, int bsad, int sad_thresh, int min_x, int min_y)
{
    if (n > limit * 2) {
        const int thr = bsad * (sad_thresh + 8) >> 4;
        int i = 0;
        for (int j = 0; j < n; j++) if (TESA_SAD(*TESA_SLOT(L, j)) <= thr) { *TESA_SLOT(L, i) = *TESA_SLOT(L, j); i++; }
        n = i;
    }
    if (n > limit) {
        for (int i = 0; i < limit; i++) {
            int bj = i, bs = TESA_SAD(*TESA_SLOT(L, i));
            for (int j = i + 1; j < n; j++) if (TESA_SAD(*TESA_SLOT(L, j)) < bs) { bs = TESA_SAD(*TESA_SLOT(L, j)); bj = j; }
            if (bj > i) { uint32_t t = *TESA_SLOT(L, i); *TESA_SLOT(L, i) = *TESA_SLOT(L, bj); *TESA_SLOT(L, bj) = t; }
        }
        n = limit;
    }
    for (int i = 0; i < n; i++) { const uint32_t e = *TESA_SLOT(L, i); L->cxy[i] = CAND_PACK((min_x + (int)(e & 63)) * 4, (min_y + (int)(e >> 6 & 63)) * 4); }
    return n;
}
static inline int prim_chroma4x4_cost(const FrameDev &F, MBLocal *L, int i8, const int mv4x[4], const int mv4y[4], int satd)
{
    int ox = 4 * (i8 & 1), oy = 2 * (i8 & 2), s = 0;
    for (int p = 0; p < 2; p++) {
        uint8_t tmp[4 * 4];
        for (int q = 0; q < 4; q++)
            emu_mc_chroma(F, tmp + (q >> 1) * 2 * 4 + (q & 1) * 2, 4, p, L->mb_x * 8 + ox + (q & 1) * 2, L->mb_y * 8 + oy + (q >> 1) * 2, mv4x[q], mv4y[q], 2, 2);
        s += emu_cmp(4, 4, L->fenc + 256 + oy * 16 + p * 8 + ox, 16, tmp, 4, satd);
    }
    return s;
}
static inline void prim_load_fenc(const FrameDev &F, MBLocal *L)
{
    for (int y = 0; y < 16; y++) memcpy(L->fenc + y * 16, F.fenc[0] + (size_t)(L->mb_y * 16 + y) * F.w + L->mb_x * 16, 16);
    for (int p = 0; p < 2; p++) for (int y = 0; y < 8; y++) memcpy(L->fenc + 256 + y * 16 + p * 8, F.fenc[1 + p] + (size_t)(L->mb_y * 8 + y) * (F.w / 2) + L->mb_x * 8, 8);
}
static inline void emu_pred_px(const FrameDev &F, MBLocal *L, int x, int y, int mvx, int mvy)
{
    uint8_t tmp[32]; int st;
    const uint8_t *r = emu_qpel(F, tmp, &st, L->mb_x * 16 + x, L->mb_y * 16 + y, mvx, mvy, 1, 1);
    L->pred[y * 16 + x] = r[0];
}
static inline void prim_win_load(const FrameDev &F, MBLocal *L, int bmx, int bmy) { (void)F; (void)L; (void)bmx; (void)bmy; }   /* the scalar prims read the planes directly */
static inline void prim_predict_mb(const FrameDev &F, MBLocal *L, int win = 0)
{
    (void)win;
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) {
        int i8 = SCAN8_0 + (x >> 2) + 8 * (y >> 2);
        emu_pred_px(F, L, x, y, clip3i(L->cmv[i8][0], L->mv_min[0], L->mv_max[0]), clip3i(L->cmv[i8][1], L->mv_min[1], L->mv_max[1]));
    }
    for (int p = 0; p < 2; p++) for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
        int i8 = SCAN8_0 + (x >> 1) + 8 * (y >> 1);
        emu_mc_chroma(F, L->pred + 256 + y * 16 + p * 8 + x, 16, p, L->mb_x * 8 + x, L->mb_y * 8 + y,
                      clip3i(L->cmv[i8][0], L->mv_min[0], L->mv_max[0]), clip3i(L->cmv[i8][1], L->mv_min[1], L->mv_max[1]), 1, 1);
    }
}
static inline void prim_predict_16x16(const FrameDev &F, MBLocal *L, int mvx, int mvy, int which)
{
    if (which != 2) for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) emu_pred_px(F, L, x, y, mvx, mvy);
    if (which != 0) for (int p = 0; p < 2; p++) emu_mc_chroma(F, L->pred + 256 + p * 8, 16, p, L->mb_x * 8, L->mb_y * 8, mvx, mvy, 8, 8);
}
static inline void prim_residual(const FrameDev &F, MBLocal *L, int do_luma, int do_chroma)
{
    emu_stats[14]++;
    for (int b = 0; b < 24; b++) {
        int is_l = b < 16;
        if (is_l ? !do_luma : !do_chroma) continue;
        int ch = (b - 16) >> 2, ci = (b - 16) & 3;
        int px = is_l ? 4 * blk_x_of(b) : ch * 8 + (ci & 1) * 4, py = is_l ? 4 * blk_y_of(b) : 16 + (ci >> 1) * 4;
        int16_t d[4][4], t[4][4], c[16];
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = L->fenc[(py + y) * 16 + px + x] - L->pred[(py + y) * 16 + px + x];
        for (int i = 0; i < 4; i++) {
            int s03 = d[i][0] + d[i][3], s12 = d[i][1] + d[i][2], d03 = d[i][0] - d[i][3], d12 = d[i][1] - d[i][2];
            t[0][i] = s03 + s12; t[1][i] = 2 * d03 + d12; t[2][i] = s03 - s12; t[3][i] = d03 - 2 * d12;
        }
        for (int i = 0; i < 4; i++) {
            int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
            c[i * 4] = s03 + s12; c[i * 4 + 1] = 2 * d03 + d12; c[i * 4 + 2] = s03 - s12; c[i * 4 + 3] = d03 - 2 * d12;
        }
        if (!is_l) { L->red[b] = c[0]; c[0] = 0; }
        int cat = is_l ? 0 : 1, qp = is_l ? F.qp : F.chroma_qp, nz = 0;
        for (int i = 0; i < 16; i++) {
            int cls = (i & 1) + ((i >> 2) & 1), mf = F.q_mf[cat][cls], bias = F.q_bias[cat][cls], v = c[i];
            v = v > 0 ? ((bias + v) * mf >> 16) : -((bias - v) * mf >> 16);
            c[i] = (int16_t)v; nz |= v;
        }
        nz = nz != 0;
        int score = 0;
        if (nz) {
            int idx = 15, lo = is_l ? 0 : 1;
            while (idx >= lo && c[zz4_tab[idx]] == 0) idx--;
            while (idx >= lo) {
                int v = c[zz4_tab[idx--]];
                if ((unsigned)(v + 1) > 2) { score = 9; break; }
                int run = 0;
                while (idx >= lo && c[zz4_tab[idx]] == 0) { idx--; run++; }
                score += decimate_tab4[run];
            }
            int qbits = qp / 6 - 4; const int *dq = is_l ? F.dq_mf : F.dq_mf_c;
            for (int i = 0; i < 16; i++) {
                int cls = (i & 1) + ((i >> 2) & 1);
                c[i] = qbits >= 0 ? (int16_t)((c[i] * dq[cls]) << qbits) : (int16_t)((c[i] * dq[cls] + (1 << (-qbits - 1))) >> (-qbits));
            }
        }
        L->blk_nz[b] = nz; L->blk_score[b] = score;
        for (int i = 0; i < 16; i++) L->coef[b][i] = c[i];
    }
    if (do_chroma)
        for (int ch = 0; ch < 2; ch++) {
            int b0 = L->red[16 + ch * 4], b1 = L->red[17 + ch * 4], b2 = L->red[18 + ch * 4], b3 = L->red[19 + ch * 4];
            int d0 = b0 + b1, d1 = b2 + b3, d2 = b0 - b1, d3 = b2 - b3;
            L->cdc[ch][0] = d0 + d1; L->cdc[ch][1] = d0 - d1; L->cdc[ch][2] = d2 + d3; L->cdc[ch][3] = d2 - d3;
        }
}
static inline void emu_idct_add(uint8_t *dst, const int16_t *c)
{
    int16_t t[4][4], r[4][4];
    for (int i = 0; i < 4; i++) {
        int s02 = c[i] + c[8 + i], d02 = c[i] - c[8 + i], s13 = c[4 + i] + (c[12 + i] >> 1), d13 = (c[4 + i] >> 1) - c[12 + i];
        t[i][0] = s02 + s13; t[i][1] = d02 + d13; t[i][2] = d02 - d13; t[i][3] = s02 - s13;
    }
    for (int i = 0; i < 4; i++) {
        int s02 = t[0][i] + t[2][i], d02 = t[0][i] - t[2][i], s13 = t[1][i] + (t[3][i] >> 1), d13 = (t[1][i] >> 1) - t[3][i];
        r[0][i] = (s02 + s13 + 32) >> 6; r[1][i] = (d02 + d13 + 32) >> 6; r[2][i] = (d02 - d13 + 32) >> 6; r[3][i] = (s02 - s13 + 32) >> 6;
    }
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * 16 + x] = (uint8_t)clip3i(dst[y * 16 + x] + r[y][x], 0, 255);
}
static inline void prim_add_idct(const FrameDev &F, MBLocal *L, unsigned keep, int cm0, int cm1)
{
    (void)F;
    for (int b = 0; b < 16; b++) if (((keep >> b) & 1) && L->blk_nz[b]) emu_idct_add(L->pred + 4 * blk_y_of(b) * 16 + 4 * blk_x_of(b), L->coef[b]);
    for (int b = 16; b < 24; b++) {
        int ch = (b - 16) >> 2, ci = (b - 16) & 3, mode = ch ? cm1 : cm0;
        uint8_t *dst = L->pred + 256 + (ci >> 1) * 4 * 16 + ch * 8 + (ci & 1) * 4;
        if (mode == 2) emu_idct_add(dst, L->coef[b]);
        else if (mode == 1) { int v = (L->cdc[ch][ci] + 32) >> 6; for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * 16 + x] = (uint8_t)clip3i(dst[y * 16 + x] + v, 0, 255); }
    }
}
/* transform stage of x264_macroblock_encode for an inter macroblock (encoder/macroblock.c:277-372, 696-753),
 * scalar restatement with the reference's own loops (the GPU primitive does this lane-parallel) */
static inline void prim_mb_transform(const FrameDev &F, MBLocal *L)
{
    prim_residual(F, L, 1, 1);
    /* luma 8x8 / MB decimation (encoder/macroblock.c:696-753) */
    unsigned keep = 0; int cbp = 0, decimate_mb = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        int dec8 = 0, any = 0;
        for (int i4 = 0; i4 < 4; i4++) {
            int idx = i8 * 4 + i4;
            if (L->blk_nz[idx]) { if (F.b_dct_decimate && dec8 < 6) dec8 += L->blk_score[idx]; any = 1; }
        }
        decimate_mb += dec8;
        if (F.b_dct_decimate) { if (dec8 >= 4) cbp |= 1 << i8; }
        else if (any) cbp |= 1 << i8;
    }
    if (F.b_dct_decimate && decimate_mb < 6) cbp = 0;
    for (int i8 = 0; i8 < 4; i8++) if (cbp & (1 << i8)) keep |= 0xFu << (4 * i8);
    L->cbp_luma = cbp;
    L->nnz_mask = 0;
    for (int idx = 0; idx < 16; idx++) if (((keep >> idx) & 1) && L->blk_nz[idx]) L->nnz_mask |= 1 << idx;
    /* chroma (encoder/macroblock.c:277-372) */
    int cmode[2], any_ac = 0;
    for (int ch = 0; ch < 2; ch++) {
        int score = 0, nz_ac = 0, nz_dc = 0;
        for (int i = 0; i < 4; i++) if (L->blk_nz[16 + ch * 4 + i]) { nz_ac = 1; if (F.b_dct_decimate) score += L->blk_score[16 + ch * 4 + i]; }
        int dc[4];
        { int mf = F.q_mf[1][0] >> 1, bias = F.q_bias[1][0] << 1;
          for (int k = 0; k < 4; k++) {
              int c = L->cdc[ch][k];
              dc[k] = c > 0 ? ((bias + c) * mf >> 16) : -((bias - c) * mf >> 16);
              nz_dc |= dc[k];
          } }
        int d0 = dc[0] + dc[1], d1 = dc[2] + dc[3], d2 = dc[0] - dc[1], d3 = dc[2] - dc[3];
        int dmf = F.dq_mf_c[0], qbits = F.chroma_qp / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        int r0 = (int16_t)((d0 + d1) * dmf >> -qbits), r1 = (int16_t)((d0 - d1) * dmf >> -qbits);
        int r2 = (int16_t)((d2 + d3) * dmf >> -qbits), r3 = (int16_t)((d2 - d3) * dmf >> -qbits);
        if ((F.b_dct_decimate && score < 7) || !nz_ac) {
            if (!nz_dc) { cmode[ch] = 0; continue; }
            cmode[ch] = 1;
            L->cdc[ch][0] = (int16_t)r0; L->cdc[ch][1] = (int16_t)r1; L->cdc[ch][2] = (int16_t)r2; L->cdc[ch][3] = (int16_t)r3;
        } else {
            any_ac = 1; cmode[ch] = 2;
            if (nz_dc) {
                L->coef[16 + ch * 4 + 0][0] = (int16_t)r0; L->coef[16 + ch * 4 + 1][0] = (int16_t)r1;
                L->coef[16 + ch * 4 + 2][0] = (int16_t)r2; L->coef[16 + ch * 4 + 3][0] = (int16_t)r3;
            }
        }
    }
    L->cbp_chroma = any_ac ? 2 : 0;
    prim_add_idct(F, L, keep, cmode[0], cmode[1]);
}
/* the four-at-once RCA re-encodes, scalar: one prediction / transform after the other */
static inline void prim_predict_win16(const FrameDev &F, MBLocal *L, int j, int mvx, int mvy)
{
    uint8_t keep[24 * 16];
    memcpy(keep, L->pred, sizeof(keep));
    prim_predict_16x16(F, L, mvx, mvy, 1);
    memcpy(L->pred4[j], L->pred, sizeof(keep));
    memcpy(L->pred, keep, sizeof(keep));
}
static inline void prim_mb_transform4(const FrameDev &F, MBLocal *L)
{
    uint8_t in[4][24 * 16], keep[24 * 16];
    memcpy(in, L->pred4, sizeof(in));          /* pred4 shares its storage with the coefficient scratch of the scalar transform */
    memcpy(keep, L->pred, sizeof(keep));
    for (int j = 0; j < 4; j++) {
        memcpy(L->pred, in[j], sizeof(keep));
        prim_mb_transform(F, L);
        memcpy(in[j], L->pred, sizeof(keep));
    }
    memcpy(L->pred, keep, sizeof(keep));
    memcpy(L->pred4, in, sizeof(in));
}
static inline int prim_chroma_ssd(const FrameDev &F, MBLocal *L, int ch)
{
    (void)F; int s = 0;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { int d = L->fenc[256 + y * 16 + ch * 8 + x] - L->pred[256 + y * 16 + ch * 8 + x]; s += d * d; }
    return s;
}
static inline void prim_copy_pred(MBLocal *L, uint8_t *dst) { memcpy(dst, L->pred, sizeof(L->recb)); }
static inline void prim_store_rec(const FrameDev &F, MBLocal *L)
{
    for (int y = 0; y < 16; y++) memcpy(F.rec[0] + (size_t)(L->mb_y * 16 + y) * F.w + L->mb_x * 16, L->pred + y * 16, 16);
    for (int p = 0; p < 2; p++) for (int y = 0; y < 8; y++) memcpy(F.rec[1 + p] + (size_t)(L->mb_y * 8 + y) * (F.w / 2) + L->mb_x * 8, L->pred + 256 + y * 16 + p * 8, 8);
}
static inline void prim_store_mvr(const FrameDev &F, MBLocal *L, int mvx, int mvy) { F.mvr[2 * L->mb_xy] = (int16_t)mvx; F.mvr[2 * L->mb_xy + 1] = (int16_t)mvy; }
#endif
