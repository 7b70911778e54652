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
#include "pcamv_entropy_tables.h"

static inline void predict_mv(MBLocal *L, int idx, int width, int mvp[2]);
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
static inline int prim_mv_cost(const FrameDev &F, int d) { return (int)F.cost_mv[d]; }       /* entry d = mv - mvp of the MV-bit table (centre pointer) */
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
static inline void prim_mb_transform(const FrameDev &F, MBLocal *L, int lv = 0)
{
    prim_residual(F, L, 1, 1);
    int16_t lvq[24][16];           /* quantised levels in scan order: requantise from the forward transform (prim_residual keeps only the dequantised ones) */
    if (lv)
        for (int b = 0; b < 24; b++) {
            int is_l = b < 16, ch = (b - 16) >> 2, ci = (b - 16) & 3;
            int px = is_l ? 4 * blk_x_of(b) : ch * 8 + (ci & 1) * 4, py = is_l ? 4 * blk_y_of(b) : 16 + (ci >> 1) * 4;
            int d[4][4], t[4][4], c[16];
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = L->fenc[(py + y) * 16 + px + x] - L->pred[(py + y) * 16 + px + x];
            for (int i = 0; i < 4; i++) {
                int s03 = d[i][0] + d[i][3], s12 = d[i][1] + d[i][2], d03 = d[i][0] - d[i][3], d12 = d[i][1] - d[i][2];
                t[0][i] = s03 + s12; t[1][i] = 2 * d03 + d12; t[2][i] = s03 - s12; t[3][i] = d03 - 2 * d12;
            }
            for (int i = 0; i < 4; i++) {
                int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
                c[i * 4] = s03 + s12; c[i * 4 + 1] = 2 * d03 + d12; c[i * 4 + 2] = s03 - s12; c[i * 4 + 3] = d03 - 2 * d12;
            }
            if (!is_l) c[0] = 0;
            for (int i = 0; i < 16; i++) {
                int cls = (i & 1) + ((i >> 2) & 1), mf = F.q_mf[is_l ? 0 : 1][cls], bias = F.q_bias[is_l ? 0 : 1][cls], v = c[i];
                c[i] = v > 0 ? ((bias + v) * mf >> 16) : -((bias - v) * mf >> 16);
            }
            for (int k = 0; k < 16; k++) lvq[b][k] = (int16_t)c[zz4_tab[k]];
        }
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
    int cmode[2], any_ac = 0, dcl[2][4] = {{0}}, dcnz[2] = {0, 0};
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
        for (int k = 0; k < 4; k++) dcl[ch][k] = dc[k];
        dcnz[ch] = nz_dc != 0;
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
    L->cbp_chroma = any_ac ? 2 : (dcnz[0] | dcnz[1]) ? 1 : 0;       /* encoder/macroblock.c:364-372 */
    prim_add_idct(F, L, keep, cmode[0], cmode[1]);
    if (lv) {
        for (int b = 0; b < 16; b++) L->nzc[scan8_all_of(b)] = (uint8_t)((L->nnz_mask >> b) & 1);
        for (int b = 16; b < 24; b++) L->nzc[scan8_all_of(b)] = (uint8_t)(cmode[(b - 16) >> 2] == 2 && L->blk_nz[b]);
        L->nzc[scan8_all_of(24)] = 0;
        for (int ch = 0; ch < 2; ch++) {
            L->nzc[scan8_all_of(25 + ch)] = (uint8_t)dcnz[ch];
            /* zigzag_scan_2x2_dc (encoder/macroblock.c:31-38): d[0][0], d[1][0], d[0][1], d[1][1] */
            L->cdc[ch][0] = (int16_t)dcl[ch][0]; L->cdc[ch][1] = (int16_t)dcl[ch][2]; L->cdc[ch][2] = (int16_t)dcl[ch][1]; L->cdc[ch][3] = (int16_t)dcl[ch][3];
        }
        for (int b = 0; b < 24; b++) for (int k = 0; k < 16; k++) L->coef[b][k] = lvq[b][k];
    }
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
        prim_mb_transform(F, L, 0);
        memcpy(in[j], L->pred, sizeof(keep));
    }
    memcpy(L->pred, keep, sizeof(keep));
    memcpy(L->pred4, in, sizeof(in));
}

/* ------------------------------------------------------------------ --subme >= 6 (scalar stand-ins of the RD primitives) */
static inline void prim_rd_load(const FrameDev &F, MBLocal *L)
{
    static const uint8_t top_pos[8] = {4, 5, 6, 7, 1, 2, 1 + 3 * 8, 2 + 3 * 8}, left_pos[8] = {3 + 8, 3 + 16, 3 + 24, 3 + 32, 0 + 8, 0 + 16, 0 + 32, 0 + 40};
    const int xy = L->mb_xy, top = xy - F.mb_w;
    memset(L->nzc, 0, sizeof(L->nzc)); memset(L->cmvd, 0, sizeof(L->cmvd)); memset(L->i4mode, -1, sizeof(L->i4mode));
    if (F.inter & PCAMV_ANALYSE_PSUB8x8) {       /* the macroblock's own entries as the macroblock coded before it left them (PCAMV_CHAIN_*) */
        const uint8_t *src = xy == 0 ? F.cabac_init : F.cabac;
        for (int b = 0; b < 24; b++) L->nzc[scan8_all_of(b)] = src[PCAMV_CHAIN_NZ + b];
        for (int b = 0; b < 16; b++) memcpy(L->cmvd[scan8_of(b)], src + PCAMV_CHAIN_MVD + 4 * b, 4);
    }
    for (int k = 0; k < 8; k++) {
        L->nzc[top_pos[k]] = (L->neighbour & NB_TOP) ? F.nb_nz[top * 16 + k] : 0x80;
        L->nzc[left_pos[k]] = (L->neighbour & NB_LEFT) ? F.nb_nz[(xy - 1) * 16 + 8 + k] : 0x80;
    }
    L->cbp_top = (L->neighbour & NB_TOP) ? F.nb_cbp[top] : -1;
    L->cbp_left = (L->neighbour & NB_LEFT) ? F.nb_cbp[xy - 1] : -1;
    for (int k = 0; k < 4; k++) {
        if (L->neighbour & NB_TOP) { L->cmvd[SCAN8_0 - 8 + k][0] = F.nb_mvd[(top * 8 + k) * 2]; L->cmvd[SCAN8_0 - 8 + k][1] = F.nb_mvd[(top * 8 + k) * 2 + 1]; L->i4mode[SCAN8_0 - 8 + k] = 2; }
        if (L->neighbour & NB_LEFT) { L->cmvd[SCAN8_0 - 1 + 8 * k][0] = F.nb_mvd[((xy - 1) * 8 + 4 + k) * 2]; L->cmvd[SCAN8_0 - 1 + 8 * k][1] = F.nb_mvd[((xy - 1) * 8 + 4 + k) * 2 + 1]; L->i4mode[SCAN8_0 - 1 + 8 * k] = 2; }
    }
    for (int c = 0; c < 3; c++) {
        const int w = c ? 8 : 16, pw = c ? F.w / 2 : F.w, x0 = L->mb_x * w, y0 = L->mb_y * w;
        for (int x = -1; x < w + w / 2; x++) L->ib_top[c][4 + x] = L->mb_y > 0 ? F.rec[c][(size_t)(y0 - 1) * pw + clip3i(x0 + x, 0, pw - 1)] : 0;
        for (int y = 0; y < w; y++) L->ib_left[c][y] = L->mb_x > 0 ? F.rec[c][(size_t)(y0 + y) * pw + x0 - 1] : 0;
    }
    if (F.b_cabac) {
        memcpy(L_CAB(L, 0), xy == 0 ? F.cabac_init : F.cabac, 464);
        memcpy(L_CTAB(L), F.cabac_tab, 1024);
    }
    L->b_fast_intra = xy > 4 && F.ref_is_inter;          /* analyse.c:363-378 */
}
struct MbFetch { int unused; };
static inline void prim_mb_fetch(const FrameDev &, int, int, int, int, MbFetch &) {}
static inline void prim_mb_fetch_store(const FrameDev &F, MBLocal *L, int rd, const MbFetch &) { prim_load_fenc(F, L); if (rd) prim_rd_load(F, L); }
#define EF1(a, b) (((a) + (b) + 1) >> 1)
#define EF2(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
/* H.264 8.3.1.2 / 8.3.3 / 8.3.4 from explicit neighbour arrays: top[-1] is the top-left sample */
static inline void emu_pred_plane(uint8_t *dst, int ds, int n, const uint8_t *top, const uint8_t *left)
{
    const int h = n / 2;
    int H = 0, V = 0;
    for (int i = 1; i <= h; i++) { H += i * (top[h - 1 + i] - top[h - 1 - i]); V += i * ((int)left[h - 1 + i] - (h - 1 - i < 0 ? top[-1] : left[h - 1 - i])); }
    const int a = 16 * (left[n - 1] + top[n - 1]), b = n == 16 ? (5 * H + 32) >> 6 : (17 * H + 16) >> 5, c = n == 16 ? (5 * V + 32) >> 6 : (17 * V + 16) >> 5;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * ds + x] = (uint8_t)clip3i((a + b * (x - (h - 1)) + c * (y - (h - 1)) + 16) >> 5, 0, 255);
}
static inline void prim_intra16_satd(const FrameDev &F, MBLocal *L, int avail)
{
    const uint8_t *top = L->ib_top[0] + 4, *left = L->ib_left[0];
    const int satd = F.subme > 1;
    for (int m = 0; m < 4; m++) {
        uint8_t p[16 * 16];
        const int need = m == 0 ? 2 : m == 1 ? 1 : m == 3 ? 3 : 0;
        if ((avail & need) != need || (m == 3 && avail != 3)) { L->ccost[m] = PCAMV_COST_MAX; continue; }
        int st = 0, sl = 0;
        for (int i = 0; i < 16; i++) { st += top[i]; sl += left[i]; }
        const int dc = avail == 3 ? (st + sl + 16) >> 5 : avail == 1 ? (sl + 8) >> 4 : avail == 2 ? (st + 8) >> 4 : 128;
        if (m == 3) emu_pred_plane(p, 16, 16, top, left);
        else for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) p[y * 16 + x] = (uint8_t)(m == 0 ? top[x] : m == 1 ? left[y] : dc);
        L->ccost[m] = emu_cmp(16, 16, p, 16, L->fenc, 16, satd);
    }
}
static inline void prim_intra8c_satd(const FrameDev &F, MBLocal *L, int avail)
{
    const int satd = F.subme > 1;
    for (int m = 0; m < 4; m++) {       /* DC, H, V, P */
        const int need = m == 1 ? 1 : m == 2 ? 2 : m == 3 ? 3 : 0;
        if ((avail & need) != need) { L->ccost[m] = PCAMV_COST_MAX; continue; }
        int cost = 0;
        for (int c = 1; c < 3; c++) {
            const uint8_t *top = L->ib_top[c] + 4, *left = L->ib_left[c];
            uint8_t p[8 * 8];
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int i = 0; i < 4; i++) { s0 += top[i]; s1 += top[4 + i]; s2 += left[i]; s3 += left[4 + i]; }
            if (m == 3) emu_pred_plane(p, 8, 8, top, left);
            else for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                int v;
                if (m == 1) v = left[y]; else if (m == 2) v = top[x];
                else if (avail == 3) v = (x < 4 && y < 4) ? (s0 + s2 + 4) >> 3 : y < 4 ? (s1 + 2) >> 2 : x < 4 ? (s3 + 2) >> 2 : (s1 + s3 + 4) >> 3;
                else if (avail == 1) v = y < 4 ? (s2 + 2) >> 2 : (s3 + 2) >> 2;
                else if (avail == 2) v = x < 4 ? (s0 + 2) >> 2 : (s1 + 2) >> 2;
                else v = 128;
                p[y * 8 + x] = (uint8_t)v;
            }
            cost += emu_cmp(8, 8, p, 8, L->fenc + 256 + (c - 1) * 8, 16, satd);
        }
        L->ccost[m] = cost;
    }
}
static inline void prim_intra4_init(MBLocal *L)
{
    for (int x = -1; x < 24; x++) IFD(L, x, -1) = L->ib_top[0][4 + x];
    for (int y = 0; y < 16; y++) IFD(L, -1, y) = L->ib_left[0][y];
}
static inline void emu_pred4(const MBLocal *L, int bx, int by, int mode, uint8_t p[16])
{
    int t[8], l[4], e[9];
    for (int i = 0; i < 8; i++) t[i] = L_IFD((MBLocal *)L)[(by - 1 + 1) * 32 + bx + i + 4];
    for (int i = 0; i < 4; i++) l[i] = L_IFD((MBLocal *)L)[(by + i + 1) * 32 + bx - 1 + 4];
    for (int i = 0; i < 4; i++) { e[3 - i] = l[i]; e[5 + i] = t[i]; }
    e[4] = L_IFD((MBLocal *)L)[(by - 1 + 1) * 32 + bx - 1 + 4];
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        int v;
        switch (mode) {
        case 0: v = t[x]; break;
        case 1: v = l[y]; break;
        case 2: v = (l[0] + l[1] + l[2] + l[3] + t[0] + t[1] + t[2] + t[3] + 4) >> 3; break;
        case 9: v = (l[0] + l[1] + l[2] + l[3] + 2) >> 2; break;
        case 10: v = (t[0] + t[1] + t[2] + t[3] + 2) >> 2; break;
        case 11: v = 128; break;
        case 3: v = (x == 3 && y == 3) ? EF2(t[6], t[7], t[7]) : EF2(t[x + y], t[x + y + 1], t[x + y + 2]); break;
        case 4: { int c = 4 + x - y; v = EF2(e[c - 1], e[c], e[c + 1]); } break;
        case 5: { int z = 2 * x - y, c = 4 + x - (y >> 1); v = z < -1 ? EF2(e[4 - y], e[5 - y], e[6 - y]) : (z & 1) ? EF2(e[c - 1], e[c], e[c + 1]) : EF1(e[c], e[c + 1]); } break;
        case 6: { int z = 2 * y - x, r = y - (x >> 1); v = z < -1 ? EF2(e[2 + x], e[3 + x], e[4 + x]) : (z & 1) ? EF2(e[5 - r], e[4 - r], e[3 - r]) : EF1(e[4 - r], e[3 - r]); } break;
        case 7: { int k = x + (y >> 1); v = (y & 1) ? EF2(t[k], t[k + 1], t[k + 2]) : EF1(t[k], t[k + 1]); } break;
        default: { int z = x + 2 * y, r = y + (x >> 1); v = z > 5 ? l[3] : z == 5 ? EF2(l[2], l[3], l[3]) : (z & 1) ? EF2(l[r], l[r + 1], l[r + 2]) : EF1(l[r], l[r + 1]); } break;
        }
        p[y * 4 + x] = (uint8_t)v;
    }
}
static inline void prim_intra4_costs(const FrameDev &F, MBLocal *L, int idx, int n, int emulate)
{
    const int bx = 4 * blk_x_of(idx), by = 4 * blk_y_of(idx);
    if (emulate) for (int i = 4; i < 8; i++) IFD(L, bx + i, by - 1) = IFD(L, bx + 3, by - 1);
    for (int i = 0; i < n; i++) {
        uint8_t p[16];
        emu_pred4(L, bx, by, L->slots[i], p);
        L->ccost[i] = emu_cmp(4, 4, p, 4, L->fenc + by * 16 + bx, 16, F.subme > 1);
    }
}
static inline void prim_intra4_encode(const FrameDev &F, MBLocal *L, int idx, int mode)
{
    const int bx = 4 * blk_x_of(idx), by = 4 * blk_y_of(idx);
    uint8_t p[16]; int16_t d[4][4], t[4][4], c[16];
    emu_pred4(L, bx, by, mode, p);
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = L->fenc[(by + y) * 16 + bx + x] - p[y * 4 + x];
    for (int i = 0; i < 4; i++) {
        int s03 = d[i][0] + d[i][3], s12 = d[i][1] + d[i][2], d03 = d[i][0] - d[i][3], d12 = d[i][1] - d[i][2];
        t[0][i] = s03 + s12; t[1][i] = 2 * d03 + d12; t[2][i] = s03 - s12; t[3][i] = d03 - 2 * d12;
    }
    for (int i = 0; i < 4; i++) {
        int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
        c[i * 4] = s03 + s12; c[i * 4 + 1] = 2 * d03 + d12; c[i * 4 + 2] = s03 - s12; c[i * 4 + 3] = d03 - 2 * d12;
    }
    int nz = 0;
    for (int i = 0; i < 16; i++) {
        int cls = (i & 1) + ((i >> 2) & 1), v = c[i];
        v = v > 0 ? ((F.q_bias_i[cls] + v) * F.q_mf_i[cls] >> 16) : -((F.q_bias_i[cls] - v) * F.q_mf_i[cls] >> 16);
        c[i] = (int16_t)v; nz |= v;
    }
    uint8_t tmp[4 * 16];
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) tmp[y * 16 + x] = p[y * 4 + x];
    L->nzc[scan8_of(idx)] = (uint8_t)(nz != 0);         /* encoder/macroblock.c:135: stays in the cache */
    if (nz) {
        int qbits = F.qp / 6 - 4;
        for (int i = 0; i < 16; i++) {
            int cls = (i & 1) + ((i >> 2) & 1);
            c[i] = qbits >= 0 ? (int16_t)((c[i] * F.dq_mf[cls]) << qbits) : (int16_t)((c[i] * F.dq_mf[cls] + (1 << (-qbits - 1))) >> (-qbits));
        }
        emu_idct_add(tmp, c);
    }
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) IFD(L, bx + x, by + y) = tmp[y * 16 + x];
}
/* sums of |Hadamard coefficients| of the 16x16 block at p (stride st): over its sixteen 4x4 transforms, over its four 8x8
 * transforms, and the pixel sum (common/pixel.c:256-358) */
static inline void emu_had_sums(const uint8_t *p, int st, int s4[16], int dc4[16], int s8[4])
{
    for (int b = 0; b < 16; b++) {
        const uint8_t zero[4] = {0};
        const int x = 4 * (b & 3), y = 4 * (b >> 2);
        s4[b] = emu_had4(p + y * st + x, st, zero, 0);
        dc4[b] = 0;
        for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) dc4[b] += p[(y + j) * st + x + i];
    }
    for (int b = 0; b < 4; b++) {
        int d[8][8];
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) d[y][x] = p[(8 * (b >> 1) + y) * st + 8 * (b & 1) + x];
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i < 8; i++) {
                int v[8];
                for (int k = 0; k < 8; k++) v[k] = pass ? d[k][i] : d[i][k];
                for (int step = 1; step < 8; step <<= 1) for (int k = 0; k < 8; k++) if (!(k & step)) { int a = v[k], c = v[k + step]; v[k] = a + c; v[k + step] = a - c; }
                for (int k = 0; k < 8; k++) { if (pass) d[k][i] = v[k]; else d[i][k] = v[k]; }
            }
        s8[b] = 0;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) s8[b] += iabs(d[y][x]);
    }
}
static inline void prim_fenc_complexity(const FrameDev &F, MBLocal *L)     /* x264_mb_cache_fenc_satd, analyse.c:522-549 */
{
    L->fenc_satd_sum = L->fenc_sa8d_sum = 0;
    if (!F.psy_rd) return;
    int s4[16], dc4[16], s8[4];
    emu_had_sums(L->fenc, 16, s4, dc4, s8);
    for (int b = 0; b < 16; b++) L->fenc_satd_sum += (s4[b] >> 1) - (dc4[b] >> 1);
    for (int b = 0; b < 4; b++) {
        const int r = 4 * (b >> 1) * 2 + 2 * (b & 1);      /* first 4x4 (raster) of 8x8 b */
        const int dc = dc4[r] + dc4[r + 1] + dc4[r + 4] + dc4[r + 5];
        L->fenc_sa8d_sum += ((s8[b] + 2) >> 2) - (dc >> 2);
    }
}
static inline int prim_ssd_mb(const FrameDev &F, MBLocal *L)                /* ssd_mb, rdo.c:106-137 */
{
    int ssd = 0;
    for (int y = 0; y < 24; y++) for (int x = 0; x < 16; x++) { int d = L->fenc[y * 16 + x] - L->pred[y * 16 + x]; ssd += d * d; }
    if (F.psy_rd) {
        int s4[16], dc4[16], s8[4], sum4 = 0, sum8 = 0, dc = 0;
        emu_had_sums(L->pred, 16, s4, dc4, s8);
        for (int b = 0; b < 16; b++) { sum4 += s4[b]; dc += dc4[b]; }
        for (int b = 0; b < 4; b++) sum8 += s8[b];
        int satd = (iabs(((sum4 - dc) >> 1) - L->fenc_satd_sum) + iabs(((sum8 - dc) >> 2) - L->fenc_sa8d_sum)) >> 1;
        ssd += (satd * F.psy_rd * F.lambda + 128) >> 8;
    }
    return ssd;
}
/* the CABAC walk of one macroblock: a trial walks a copy of the slice's states, the committing walk the states themselves */
struct CabWalk { uint8_t tmp[464]; uint8_t *S; int bits; };
static inline void prim_cab_begin(MBLocal *L, CabWalk &C, int trial) { (void)trial; memcpy(C.tmp, L_CAB(L, 0), 464); C.S = C.tmp; C.bits = 0; }
static inline void prim_cb_dec(MBLocal *L, CabWalk &C, int ctx, int b)
{
    const uint32_t w = L_CTAB(L)[2 * C.S[ctx] + b];
    C.S[ctx] = (uint8_t)(w & 255u); C.bits += (int)(w >> 8);
}
static inline void prim_cb_bypass(CabWalk &C, int f8) { C.bits += f8; }
static inline void prim_cb_run(MBLocal *L, CabWalk &C, int ctx, int ones, int zero)
{
    for (int i = 0; i < ones; i++) prim_cb_dec(L, C, ctx, 1);
    if (zero) prim_cb_dec(L, C, ctx, 0);
}
static inline int prim_cab_end(MBLocal *L, CabWalk &C, int commit) { if (commit) memcpy(L_CAB(L, 0), C.S, 464); else memcpy(L_CABT(L), C.S, PCAMV_CAB_USED); return C.bits; }
static inline void prim_rd_keep(const FrameDev &F, MBLocal *L)
{
    memcpy(L->snap_pred, L->pred, 384); memcpy(L->snap_nzc, L->nzc, 48); memcpy(L->snap_cmvd, L->cmvd, sizeof(L->cmvd));
    if (F.b_cabac) memcpy(L_CABK(L), L_CABT(L), PCAMV_CAB_USED);
    L->snap_cbp_luma = L->cbp_luma; L->snap_cbp_chroma = L->cbp_chroma; L->snap_nnz_mask = L->nnz_mask;
}
static inline void prim_rd_restore(const FrameDev &F, MBLocal *L)
{
    memcpy(L->pred, L->snap_pred, 384); memcpy(L->nzc, L->snap_nzc, 48); memcpy(L->cmvd, L->snap_cmvd, sizeof(L->cmvd));
    if (F.b_cabac) memcpy(L_CAB(L, 0), L_CABK(L), PCAMV_CAB_USED);
    L->cbp_luma = L->snap_cbp_luma; L->cbp_chroma = L->snap_cbp_chroma; L->nnz_mask = L->snap_nnz_mask;
}
/* residual_block_cabac for every coded block of the macroblock (encoder/cabac.c:582-667, 1000-1018) */
static inline void prim_cab_residual(const FrameDev &F, MBLocal *L, CabWalk &C, int commit)
{
    static const uint16_t sig_off[5] = {105, 120, 134, 149, 152}, last_off[5] = {166, 181, 195, 210, 213}, lvl_off[5] = {227, 237, 247, 257, 266};
    static const uint8_t lvl1[8] = {1, 2, 3, 4, 0, 0, 0, 0}, lvlgt1[8] = {5, 5, 5, 5, 6, 7, 8, 9}, nxt[2][8] = {{1, 2, 3, 3, 4, 5, 6, 7}, {4, 4, 4, 4, 5, 6, 7, 7}};
    (void)F; (void)commit;
    if (!(L->cbp_luma | L->cbp_chroma)) return;
    for (int pass = 0; pass < 3; pass++) {
        const int cat = 2 + pass, first = pass == 0 ? 0 : pass == 1 ? 25 : 16, nb = pass == 0 ? 16 : pass == 1 ? 2 : 8, count = pass == 0 ? 16 : pass == 1 ? 4 : 15;
        if (pass == 1 && !(L->cbp_chroma & 3)) continue;
        if (pass == 2 && !(L->cbp_chroma & 2)) continue;
        for (int k = 0; k < nb; k++) {
            const int idx = first + k;
            if (pass == 0 && !(L->cbp_luma & (1 << (k >> 2)))) continue;
            int inc;
            if (cat == 3) {
                const int a = L->cbp_left != -1 ? (L->cbp_left >> (9 + k)) & 1 : 0, b = L->cbp_top != -1 ? (L->cbp_top >> (9 + k)) & 1 : 0;
                inc = 4 * cat + 2 * b + a;
            } else {
                const int a = L->nzc[scan8_all_of(idx) - 1] & 0x7f, b = L->nzc[scan8_all_of(idx) - 8] & 0x7f;
                inc = 4 * cat + 2 * !!b + !!a;
            }
            const int16_t *l = pass == 1 ? L->cdc[k] : pass == 2 ? L->coef[idx] + 1 : L->coef[idx];
            const int flag = L->nzc[scan8_all_of(idx)] != 0;
            prim_cb_dec(L, C, 85 + inc, flag);
            if (!flag) continue;
            int last = count - 1;
            while (last >= 0 && !l[last]) last--;
            for (int i = 0; i < imin(last + 1, count - 1); i++) {
                prim_cb_dec(L, C, sig_off[cat] + i, l[i] != 0);
                if (l[i]) prim_cb_dec(L, C, last_off[cat] + i, i == last);
            }
            int node = 0;
            for (int i = last; i >= 0; i--) {
                if (!l[i]) continue;
                const int am1 = iabs(l[i]) - 1, prefix = imin(am1, 14);
                if (prefix) {
                    prim_cb_dec(L, C, lvl_off[cat] + lvl1[node], 1);
                    for (int q = 0; q < prefix - 1; q++) prim_cb_dec(L, C, lvl_off[cat] + lvlgt1[node], 1);
                    if (prefix < 14) prim_cb_dec(L, C, lvl_off[cat] + lvlgt1[node], 0); else C.bits += size_ue_of((unsigned)(am1 - 14)) << 8;
                    node = nxt[1][node];
                } else { prim_cb_dec(L, C, lvl_off[cat] + lvl1[node], 0); node = nxt[0][node]; }
                C.bits += 256;
            }
        }
    }
}
static inline int emu_cavlc_level(int level, int *suffix_len)
{
    int sl = *suffix_len, a = iabs(level), code = a * 2 - 2 + (level < 0), size, next = sl;
    if ((code >> sl) < 14) size = (code >> sl) + 1 + sl;
    else if (sl == 0 && code < 30) size = 19;
    else if (sl > 0 && (code >> sl) == 14) size = 15 + sl;
    else { code -= 15 << sl; if (sl == 0) code -= 15; size = 28; if (code >= 1 << 12) size += 1000000; }
    if (next == 0) next++;
    if (a > (3 << (next - 1)) && next < 6) next++;
    *suffix_len = next;
    return size;
}
static inline int emu_cavlc_block(MBLocal *L, int idx, const int16_t *l, int count)
{
    static const uint8_t ct_index[17] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3};
    int nC = 4, bits = 0;
    if (idx < 25) { int r = L->nzc[scan8_all_of(idx) - 1] + L->nzc[scan8_all_of(idx) - 8]; if (r < 0x80) r = (r + 1) >> 1; nC = ct_index[r & 0x7f]; }
    if (!L->nzc[scan8_all_of(idx)]) return pcamv_vlc_coeff0_len[nC];
    int level[16], run[16], last = count - 1, total = 0;
    while (last >= 0 && !l[last]) last--;
    for (int i = last; i >= 0;) { int r = 0; level[total] = l[i]; while (--i >= 0 && !l[i]) r++; run[total++] = r; }
    int total_zero = last + 1 - total, trailing = 0;
    L->nzc[scan8_all_of(idx)] = (uint8_t)total;
    while (trailing < 3 && trailing < total && iabs(level[trailing]) == 1) trailing++;
    bits += pcamv_vlc_coeff_len[nC * 64 + total * 4 + trailing - 4] + trailing;
    int sl = total > 10 && trailing < 3;
    if (trailing < total) {
        int v = level[trailing], s1 = sl, s2 = sl;
        if (trailing < 3) v -= v < 0 ? -1 : 1;
        bits += emu_cavlc_level(v, &s1);
        emu_cavlc_level(level[trailing], &s2); sl = s2;
        for (int i = trailing + 1; i < total; i++) bits += emu_cavlc_level(level[i], &sl);
    }
    if (total < count) bits += idx >= 25 ? pcamv_vlc_total_zeros_dc_len[(total - 1) * 4 + total_zero] : pcamv_vlc_total_zeros_len[(total - 1) * 16 + total_zero];
    for (int i = 0; i < total - 1 && total_zero > 0; i++) { bits += pcamv_vlc_run_before_len[imin(total_zero - 1, 6) * 16 + run[i]]; total_zero -= run[i]; }
    return bits;
}
static inline int emu_size_se(int v) { return size_ue_of((unsigned)(v <= 0 ? -v * 2 : v * 2 - 1)); }
/* x264_macroblock_write_cavlc as a bit counter (encoder/cavlc.c:290-600, rdo.c:41-47): leaves the coefficient counts in L->nzc */
static inline int prim_cavlc_mb(const FrameDev &F, MBLocal *L)
{
    static const uint8_t cbp_golomb[48] = {0, 2, 3, 7, 4, 8, 17, 13, 5, 18, 9, 14, 10, 15, 16, 11, 1, 32, 33, 36, 34, 37, 44, 40,
                                           35, 45, 38, 41, 39, 42, 43, 19, 6, 24, 25, 20, 26, 21, 46, 28, 27, 47, 22, 29, 23, 30, 31, 12};
    static const uint8_t sub_golomb[4] = {3, 1, 2, 0};
    int bits = 0, mvp[2];
    (void)F;
#define EMVD(idx, w) (predict_mv(L, idx, w, mvp), emu_size_se(L->cmv[scan8_of(idx)][0] - mvp[0]) + emu_size_se(L->cmv[scan8_of(idx)][1] - mvp[1]))
    if (L->i_type == PCAMV_P_8x8) {
        bits += size_ue_of(3);
        for (int i = 0; i < 4; i++) bits += size_ue_of(sub_golomb[L->sub_part[i]]);
        for (int i = 0; i < 4; i++)
            switch (L->sub_part[i]) {
            case PCAMV_D_L0_8x8: bits += EMVD(4 * i, 2); break;
            case PCAMV_D_L0_8x4: bits += EMVD(4 * i, 2); bits += EMVD(4 * i + 2, 2); break;
            case PCAMV_D_L0_4x8: bits += EMVD(4 * i, 1); bits += EMVD(4 * i + 1, 1); break;
            default: for (int k = 0; k < 4; k++) bits += EMVD(4 * i + k, 1); break;
            }
    } else if (L->i_partition == PCAMV_D_16x16) { bits += size_ue_of(0); bits += EMVD(0, 4); }
    else if (L->i_partition == PCAMV_D_16x8) { bits += size_ue_of(1); bits += EMVD(0, 4); bits += EMVD(8, 4); }
    else { bits += size_ue_of(2); bits += EMVD(0, 2); bits += EMVD(4, 2); }
#undef EMVD
    bits += size_ue_of(cbp_golomb[(L->cbp_chroma << 4) | L->cbp_luma]);
    if (L->cbp_luma | L->cbp_chroma) {
        bits += 1;
        for (int i = 0; i < 16; i++) if (L->cbp_luma & (1 << (i / 4))) bits += emu_cavlc_block(L, i, L->coef[i], 16);
    }
    if (L->cbp_chroma) {
        bits += emu_cavlc_block(L, 25, L->cdc[0], 4) + emu_cavlc_block(L, 26, L->cdc[1], 4);
        if (L->cbp_chroma & 2) for (int i = 16; i < 24; i++) bits += emu_cavlc_block(L, i, L->coef[i] + 1, 15);
    }
    return bits;
}
/* ---- x264_rd_cost_part for one 8x8 of a P_8x8 macroblock (pcamv_logic.h rd_cost_part8) ---- */
/* x264_macroblock_encode_p8x8 (encoder/macroblock.c:929-1052): prediction with the sub-partition's MVs, the 8x8's four luma blocks
 * with their own decimation rule, its two chroma 4x4 blocks without DC; levels, non-zero flags and reconstruction stay */
static inline void prim_encode_p8x8(const FrameDev &F, MBLocal *L, int i8)
{
    prim_predict_mb(F, L, 0);                   /* (the whole macroblock: only this 8x8's pixels are looked at) */
    int dec8 = 0, nnz8x8 = 0, blk_nz[4];
    int16_t deq[6][16];
    for (int k = 0; k < 6; k++) {
        const int is_l = k < 4, b = is_l ? 4 * i8 + k : 16 + i8 + (k - 4) * 4;
        const int px = is_l ? 4 * blk_x_of(b) : (k - 4) * 8 + (i8 & 1) * 4, py = is_l ? 4 * blk_y_of(b) : 16 + (i8 >> 1) * 4;
        int d[4][4], t[4][4], c[16];
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = L->fenc[(py + y) * 16 + px + x] - L->pred[(py + y) * 16 + px + x];
        for (int i = 0; i < 4; i++) {
            int s03 = d[i][0] + d[i][3], s12 = d[i][1] + d[i][2], d03 = d[i][0] - d[i][3], d12 = d[i][1] - d[i][2];
            t[0][i] = s03 + s12; t[1][i] = 2 * d03 + d12; t[2][i] = s03 - s12; t[3][i] = d03 - 2 * d12;
        }
        for (int i = 0; i < 4; i++) {
            int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
            c[i * 4] = s03 + s12; c[i * 4 + 1] = 2 * d03 + d12; c[i * 4 + 2] = s03 - s12; c[i * 4 + 3] = d03 - 2 * d12;
        }
        if (!is_l) c[0] = 0;
        int nz = 0;
        for (int i = 0; i < 16; i++) {
            int cls = (i & 1) + ((i >> 2) & 1), mf = F.q_mf[is_l ? 0 : 1][cls], bias = F.q_bias[is_l ? 0 : 1][cls], v = c[i];
            c[i] = v > 0 ? ((bias + v) * mf >> 16) : -((bias - v) * mf >> 16);
            nz |= c[i];
        }
        nz = nz != 0;
        L->nzc[scan8_all_of(b)] = (uint8_t)nz;
        if (is_l) blk_nz[k] = nz;
        if (nz) {
            int16_t lv[16];
            for (int q = 0; q < 16; q++) L->coef[b][q] = lv[q] = (int16_t)c[zz4_tab[q]];
            if (is_l) {
                nnz8x8 = 1;
                if (F.b_dct_decimate) {         /* x264_decimate_score16 */
                    int idx = 15, score = 0;
                    while (idx >= 0 && lv[idx] == 0) idx--;
                    while (idx >= 0) {
                        if ((unsigned)(lv[idx--] + 1) > 2) { score = 9; break; }
                        int run = 0;
                        while (idx >= 0 && lv[idx] == 0) { idx--; run++; }
                        score += decimate_tab4[run];
                    }
                    dec8 += score;
                }
            }
            const int qp = is_l ? F.qp : F.chroma_qp, qbits = qp / 6 - 4;
            for (int i = 0; i < 16; i++) {
                int cls = (i & 1) + ((i >> 2) & 1), dq = is_l ? F.dq_mf[cls] : F.dq_mf_c[cls];
                deq[k][i] = qbits >= 0 ? (int16_t)((c[i] * dq) << qbits) : (int16_t)((c[i] * dq + (1 << (-qbits - 1))) >> (-qbits));
            }
            if (!is_l) emu_idct_add(L->pred + py * 16 + px, deq[k]);
        }
    }
    if (F.b_dct_decimate && dec8 < 4) nnz8x8 = 0;
    for (int k = 0; k < 4; k++) {
        const int b = 4 * i8 + k;
        if (nnz8x8) { if (blk_nz[k]) emu_idct_add(L->pred + 4 * blk_y_of(b) * 16 + 4 * blk_x_of(b), deq[k]); }
        else L->nzc[scan8_of(b)] = 0;            /* STORE_8x8_NNZ( i8, 0 ) */
    }
    L->cbp_luma = (L->cbp_luma & ~(1 << i8)) | nnz8x8 << i8;
    L->cbp_chroma = 2;
}
/* ssd_plane( PIXEL_8x8, luma ) + ssd_plane( PIXEL_4x4, U / V ) of that 8x8 (rdo.c:106-128) */
static inline int prim_ssd_part8(const FrameDev &F, MBLocal *L, int i8)
{
    const int x8 = (i8 & 1) * 8, y8 = (i8 >> 1) * 8;
    int ssd = 0;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { int d = L->fenc[(y8 + y) * 16 + x8 + x] - L->pred[(y8 + y) * 16 + x8 + x]; ssd += d * d; }
    for (int ch = 0; ch < 2; ch++)
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
            const int o = (16 + y8 / 2 + y) * 16 + ch * 8 + x8 / 2 + x, d = L->fenc[o] - L->pred[o];
            ssd += d * d;
        }
    if (F.psy_rd) {
        int s4[16], dc4[16], s8[4], f4[16], fdc4[16], f8[4];
        emu_had_sums(L->pred, 16, s4, dc4, s8);
        emu_had_sums(L->fenc, 16, f4, fdc4, f8);
        const int r = 4 * (i8 >> 1) * 2 + 2 * (i8 & 1);      /* first 4x4 (raster) of the 8x8 */
        const int rr[4] = {r, r + 1, r + 4, r + 5};
        int sum4 = 0, dc = 0, fsatd = 0, fdc = 0;
        for (int k = 0; k < 4; k++) { sum4 += s4[rr[k]]; dc += dc4[rr[k]]; fsatd += (f4[rr[k]] >> 1) - (fdc4[rr[k]] >> 1); fdc += fdc4[rr[k]]; }
        const int fsa8d = ((f8[i8] + 2) >> 2) - (fdc >> 2);
        const int satd = (iabs(((sum4 - dc) >> 1) - fsatd) + iabs(((s8[i8] - dc) >> 2) - fsa8d)) >> 1;
        ssd += (satd * F.psy_rd * F.lambda + 128) >> 8;
    }
    return ssd;
}
/* the residual part of x264_partition_size_cabac (encoder/cabac.c:1058-1074) for the 8x8: its four luma blocks when the 8x8 is
 * coded, then its two chroma AC blocks (their coded_block_flags are always written) */
static inline void emu_cab_block(MBLocal *L, CabWalk &C, int cat, int idx, const int16_t *l, int count)
{
    static const uint16_t sig_off[5] = {105, 120, 134, 149, 152}, last_off[5] = {166, 181, 195, 210, 213}, lvl_off[5] = {227, 237, 247, 257, 266};
    static const uint8_t lvl1[8] = {1, 2, 3, 4, 0, 0, 0, 0}, lvlgt1[8] = {5, 5, 5, 5, 6, 7, 8, 9}, nxt[2][8] = {{1, 2, 3, 3, 4, 5, 6, 7}, {4, 4, 4, 4, 5, 6, 7, 7}};
    const int a = L->nzc[scan8_all_of(idx) - 1] & 0x7f, b = L->nzc[scan8_all_of(idx) - 8] & 0x7f;
    const int inc = 4 * cat + 2 * !!b + !!a, flag = L->nzc[scan8_all_of(idx)] != 0;
    prim_cb_dec(L, C, 85 + inc, flag);
    if (!flag) return;
    int last = count - 1;
    while (last >= 0 && !l[last]) last--;
    for (int i = 0; i < imin(last + 1, count - 1); i++) {
        prim_cb_dec(L, C, sig_off[cat] + i, l[i] != 0);
        if (l[i]) prim_cb_dec(L, C, last_off[cat] + i, i == last);
    }
    int node = 0;
    for (int i = last; i >= 0; i--) {
        if (!l[i]) continue;
        const int am1 = iabs(l[i]) - 1, prefix = imin(am1, 14);
        if (prefix) {
            prim_cb_dec(L, C, lvl_off[cat] + lvl1[node], 1);
            for (int q = 0; q < prefix - 1; q++) prim_cb_dec(L, C, lvl_off[cat] + lvlgt1[node], 1);
            if (prefix < 14) prim_cb_dec(L, C, lvl_off[cat] + lvlgt1[node], 0); else C.bits += size_ue_of((unsigned)(am1 - 14)) << 8;
            node = nxt[1][node];
        } else { prim_cb_dec(L, C, lvl_off[cat] + lvl1[node], 0); node = nxt[0][node]; }
        C.bits += 256;
    }
}
static inline void prim_cab_residual_part(const FrameDev &F, MBLocal *L, CabWalk &C, int i8)
{
    (void)F;
    if (L->cbp_luma & (1 << i8)) for (int k = 0; k < 4; k++) emu_cab_block(L, C, 2, 4 * i8 + k, L->coef[4 * i8 + k], 16);
    emu_cab_block(L, C, 4, 16 + i8, L->coef[16 + i8] + 1, 15);
    emu_cab_block(L, C, 4, 20 + i8, L->coef[20 + i8] + 1, 15);
}
/* x264_partition_size_cavlc for the 8x8 (encoder/cavlc.c:621-661): the sub-partition's MV differences, its luma blocks when coded,
 * its two chroma AC blocks; leaves the coefficient counts in L->nzc */
static inline int prim_cavlc_part8(const FrameDev &F, MBLocal *L, int i8)
{
    int bits = 0, mvp[2];
    (void)F;
#define EMVD(idx, w) (predict_mv(L, idx, w, mvp), emu_size_se(L->cmv[scan8_of(idx)][0] - mvp[0]) + emu_size_se(L->cmv[scan8_of(idx)][1] - mvp[1]))
    switch (L->sub_part[i8]) {
    case PCAMV_D_L0_8x8: bits += EMVD(4 * i8, 2); break;
    case PCAMV_D_L0_8x4: bits += EMVD(4 * i8, 2); bits += EMVD(4 * i8 + 2, 2); break;
    case PCAMV_D_L0_4x8: bits += EMVD(4 * i8, 1); bits += EMVD(4 * i8 + 1, 1); break;
    default: for (int k = 0; k < 4; k++) bits += EMVD(4 * i8 + k, 1); break;
    }
#undef EMVD
    if (L->cbp_luma & (1 << i8)) for (int k = 0; k < 4; k++) bits += emu_cavlc_block(L, 4 * i8 + k, L->coef[4 * i8 + k], 16);
    bits += emu_cavlc_block(L, 16 + i8, L->coef[16 + i8] + 1, 15) + emu_cavlc_block(L, 20 + i8, L->coef[20 + i8] + 1, 15);
    return bits;
}
static inline void prim_rd_commit(const FrameDev &F, MBLocal *L, int skip)
{
    static const uint8_t bottom[8] = {10, 11, 14, 15, 18, 19, 22, 23}, right[8] = {5, 7, 13, 15, 17, 19, 21, 23};
    const int xy = L->mb_xy;
    for (int k = 0; k < 8; k++) {
        F.nb_nz[xy * 16 + k] = skip ? 0 : L->nzc[scan8_all_of(bottom[k])];
        F.nb_nz[xy * 16 + 8 + k] = skip ? 0 : L->nzc[scan8_all_of(right[k])];
    }
    F.nb_cbp[xy] = skip ? 0 : (int16_t)((F.b_cabac ? (L->nzc[scan8_all_of(25)] << 9 | L->nzc[scan8_all_of(26)] << 10) : 0) | L->cbp_chroma << 4 | L->cbp_luma);
    for (int k = 0; k < 4; k++) {
        const int16_t *b = L->cmvd[SCAN8_0 + k + 8 * 3], *r = L->cmvd[SCAN8_0 + 3 + 8 * k];
        F.nb_mvd[(xy * 8 + k) * 2] = skip ? 0 : b[0]; F.nb_mvd[(xy * 8 + k) * 2 + 1] = skip ? 0 : b[1];
        F.nb_mvd[(xy * 8 + 4 + k) * 2] = skip ? 0 : r[0]; F.nb_mvd[(xy * 8 + 4 + k) * 2 + 1] = skip ? 0 : r[1];
    }
    if (F.b_cabac) {
        memcpy(F.cabac, L_CAB(L, 0), 464);
        if (F.dbg_hash) { uint32_t h = 2166136261u; for (int i = 0; i < 460; i++) h = (h ^ L_CAB(L, 0)[i]) * 16777619u; F.dbg_hash[xy] = h; }
    }
    if (F.inter & PCAMV_ANALYSE_PSUB8x8) {
        for (int b = 0; b < 24; b++) F.cabac[PCAMV_CHAIN_NZ + b] = skip ? 0 : L->nzc[scan8_all_of(b)];
        for (int b = 0; b < 16; b++) memcpy(F.cabac + PCAMV_CHAIN_MVD + 4 * b, L->cmvd[scan8_of(b)], 4);
    }
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
