/*
 * pcamv_oracle.c -- TEST INFRASTRUCTURE ONLY (see pcamv_oracle.h).
 *
 * Scalar C restatement of the reference's pass-1 P-frame path.  Every function cites the
 * reference file:line whose behaviour it restates; the code itself is written from the
 * algorithm, organised per frame rather than around x264_t.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pcamv_oracle.h"

#define COST_MAX (1 << 28)
#define PAD 32
#define CPAD 16
#define ALIGN16(x) (((x) + 15) & ~15)
#define MIN2(a, b) ((a) < (b) ? (a) : (b))
#define MAX2(a, b) ((a) > (b) ? (a) : (b))
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline uint8_t clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : (uint8_t)v; }
static inline int median3(int a, int b, int c)
{
    int t = (a - b) & ((a - b) >> 31); a -= t; b += t;
    b -= (b - c) & ((b - c) >> 31);
    b += (a - b) & ((a - b) >> 31);
    return b;
}

enum { PIX_16x16, PIX_16x8, PIX_8x16, PIX_8x8, PIX_8x4, PIX_4x8, PIX_4x4 };
static const int pix_w[7] = {16, 16, 8, 8, 8, 4, 4};
static const int pix_h[7] = {16, 8, 16, 8, 4, 8, 4};
static const uint8_t blk_x[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3};   /* common/macroblock.h:195 */
static const uint8_t blk_y[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};
#define SCAN8_0 (4 + 1 * 8)
static inline int scan8(int idx) { return SCAN8_0 + blk_x[idx] + 8 * blk_y[idx]; }   /* common/common.h:217 */
/* common/common.h:217-238: position of block idx in the 6x8 neighbour caches (luma 0..15, Cb 16..19, Cr 20..23, luma DC, Cb DC, Cr DC) */
static const uint8_t scan8_all[27] = {
    4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
    4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8,
    1 + 1 * 8, 2 + 1 * 8, 1 + 2 * 8, 2 + 2 * 8, 1 + 4 * 8, 2 + 4 * 8, 1 + 5 * 8, 2 + 5 * 8,
    4 + 5 * 8, 5 + 5 * 8, 6 + 5 * 8};

/* analyse.c:148-156 */
static const int lambda_tab[52] = {
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
    6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
/* analyse.c:159-167 */
static const int lambda2_tab[52] = {
    14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322,
    2925, 3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628,
    117964, 148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
/* H.264 table 8-15 */
static const uint8_t chroma_qp_tab[52] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

/* ------------------------------------------------------------------------------------------
 * pixel metrics: common/pixel.c:40-65 (SAD), :71-96 (SSD), :187-253 (SATD)
 * ---------------------------------------------------------------------------------------- */
int orc_sad(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb)
{
    int s = 0;
    for (int y = 0; y < pix_h[i_pixel]; y++, a += sa, b += sb)
        for (int x = 0; x < pix_w[i_pixel]; x++) s += abs(a[x] - b[x]);
    return s;
}
int orc_ssd(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb)
{
    int s = 0;
    for (int y = 0; y < pix_h[i_pixel]; y++, a += sa, b += sb)
        for (int x = 0; x < pix_w[i_pixel]; x++) { int d = a[x] - b[x]; s += d * d; }
    return s;
}
/* sum of |H4 * D * H4| over one 4x4 block, not yet halved */
static int hadamard4x4_abs(const uint8_t *a, int sa, const uint8_t *b, int sb)
{
    int d[4][4], t[4][4], s = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) d[y][x] = a[y * sa + x] - b[y * sb + x];
    for (int y = 0; y < 4; y++) {
        int s01 = d[y][0] + d[y][1], d01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], d23 = d[y][2] - d[y][3];
        t[y][0] = s01 + s23; t[y][1] = d01 + d23; t[y][2] = s01 - s23; t[y][3] = d01 - d23;
    }
    for (int x = 0; x < 4; x++) {
        int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
        s += abs(s01 + s23) + abs(d01 + d23) + abs(s01 - s23) + abs(d01 - d23);
    }
    return s;
}
/* pixel.c:187-253: 4x4 alone halves its own sum; every other size is built from 8x4 units
 * (two 4x4 sums added, then halved once) except 4x8 = two halved 4x4. */
int orc_satd(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb)
{
    int w = pix_w[i_pixel], h = pix_h[i_pixel], s = 0;
    if (w == 4) {
        for (int y = 0; y < h; y += 4) s += hadamard4x4_abs(a + y * sa, sa, b + y * sb, sb) >> 1;
        return s;
    }
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 8)
            s += (hadamard4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) +
                  hadamard4x4_abs(a + y * sa + x + 4, sa, b + y * sb + x + 4, sb)) >> 1;
    return s;
}

/* ------------------------------------------------------------------------------------------
 * motion compensation: common/mc.c:194-277
 * ---------------------------------------------------------------------------------------- */
static const int hpel_ref0[16] = {0, 1, 1, 1, 0, 1, 1, 1, 2, 3, 3, 3, 0, 1, 1, 1};
static const int hpel_ref1[16] = {0, 0, 0, 0, 2, 2, 3, 2, 2, 2, 3, 2, 2, 2, 3, 2};

/* mc.c:220-243 get_ref: returns either a pointer into a plane (stride updated) or dst */
static const uint8_t *get_ref(uint8_t *dst, int *dstride, uint8_t *const src[4], int ss, int mvx, int mvy, int w, int h)
{
    int qidx = ((mvy & 3) << 2) + (mvx & 3);
    int off = (mvy >> 2) * ss + (mvx >> 2);
    const uint8_t *s1 = src[hpel_ref0[qidx]] + off + ((mvy & 3) == 3) * ss;
    if (qidx & 5) {
        const uint8_t *s2 = src[hpel_ref1[qidx]] + off + ((mvx & 3) == 3);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) dst[y * *dstride + x] = (s1[y * ss + x] + s2[y * ss + x] + 1) >> 1;
        return dst;
    }
    *dstride = ss;
    return s1;
}
/* mc.c:197-216 */
void orc_mc_luma(uint8_t *dst, int ds, uint8_t *const src[4], int ss, int mvx, int mvy, int w, int h)
{
    int st = ds;
    const uint8_t *p = get_ref(dst, &st, src, ss, mvx, mvy, w, h);
    if (p != dst)
        for (int y = 0; y < h; y++) memcpy(dst + y * ds, p + y * st, w);
}
/* mc.c:246-277 */
void orc_mc_chroma(uint8_t *dst, int ds, const uint8_t *src, int ss, int mvx, int mvy, int w, int h)
{
    int dx = mvx & 7, dy = mvy & 7;
    int cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
    src += (mvy >> 3) * ss + (mvx >> 3);
    for (int y = 0; y < h; y++, dst += ds, src += ss)
        for (int x = 0; x < w; x++)
            dst[x] = (cA * src[x] + cB * src[x + 1] + cC * src[x + ss] + cD * src[x + ss + 1] + 32) >> 6;
}

/* ------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------- */
struct orc {
    pcamv_params_t p;
    int mb_w, mb_h, n_mb;
    int stride, lines, cstride, clines;
    uint8_t *lbuf, *luma[4];
    uint8_t *cbuf[2], *chroma[2];
    uint16_t *ibuf, *integral;
    uint8_t *fenc[3], *frec[3];
    int8_t *mb_type;              /* [n_mb]                 */
    int16_t (*mv)[2];             /* [mb_h*4][mb_w*4]       */
    int8_t *ref8;                 /* [mb_h*2][mb_w*2]       */
    int16_t (*mvr)[2];            /* [n_mb] 16x16 results   */
    int16_t (*prev_mv)[2];
    int8_t *prev_ref;
    int have_prev, ref_is_inter;
    int16_t *cost_mv[52];         /* centre pointers        */
    uint16_t *cost_mv_fpel[52][4];
    uint16_t quant_mf[3][52][16], quant_bias[3][52][16];   /* [0]=inter luma (CQM_4PY), [1]=inter chroma (CQM_4PC), [2]=intra luma (CQM_4IY) */
    int dequant_mf[6][16];
    orc_rand_t rnd;
    int16_t *scratch;
    /* --subme >= 6: what the entropy coder leaves behind macroblock after macroblock (pcamv_oracle_rd.inc) */
    uint8_t cabac_state[460];
    uint8_t (*nnz)[24];           /* [n_mb] luma 0..15 (x264 block order), Cb 16..19, Cr 20..23: non-zero flags (CABAC) / counts (CAVLC) */
    int16_t *cbp;                 /* [n_mb] h->mb.cbp: luma | chroma << 4 | DC bits << 8 */
    int16_t (*mvd)[2];            /* [mb_h*4][mb_w*4] */
    uint32_t *dbg_state_hash; int dbg_mb; uint8_t *dbg_state;
};

/* analyse.c:193-209: lambda * (2*log2(i+1) + 0.718 + !!i) + .5, with the reference's own
 * log2f macro (analyse.c:46): float(log(x)) / log(2) evaluated in double. */
void orc_cost_mv_table(int qp, int16_t *out)
{
    int lambda = lambda_tab[qp];
    int16_t *c = out + 2 * 4 * 2048;
    for (int i = 0; i <= 2 * 4 * 2048; i++)
        c[-i] = c[i] = lambda * (((float)log((double)(i + 1))) / (log((double)2)) * 2 + 0.718f + !!i) + .5f;
}

/* common/set.c:68-174 for the flat matrices */
static void build_quant_tables(orc_t *o)
{
    static const int deq[6][3] = {{10, 13, 16}, {11, 14, 18}, {13, 16, 20}, {14, 18, 23}, {16, 20, 25}, {18, 23, 29}};
    static const int qnt[6][3] = {{13107, 8066, 5243}, {11916, 7490, 4660}, {10082, 6554, 4194},
                                  {9362, 5825, 3647},  {8192, 5243, 3355},  {7282, 4559, 2893}};
    /* set.c:77-79: deadzone[CQM_4PY] = 32 - i_luma_deadzone[0], deadzone[CQM_4PC] = 32 - 21 */
    int dz[3] = {32 - o->p.i_luma_deadzone[0], 32 - 21, 32 - o->p.i_luma_deadzone[1]};
    for (int q = 0; q < 6; q++)
        for (int i = 0; i < 16; i++) o->dequant_mf[q][i] = deq[q][(i & 1) + ((i >> 2) & 1)] * 16;
    for (int cat = 0; cat < 3; cat++)
        for (int q = 0; q < 52; q++)
            for (int i = 0; i < 16; i++) {
                int base = qnt[q % 6][(i & 1) + ((i >> 2) & 1)];   /* DIV(x*16,16) == x */
                int s = q / 6 - 1, j;
                j = s < 0 ? base << -s : s == 0 ? base : (base + (1 << (s - 1))) >> s;
                o->quant_mf[cat][q][i] = j;
                int b1 = ((dz[cat] << 10) + (j >> 1)) / j, b2 = (1 << 15) / j;
                o->quant_bias[cat][q][i] = MIN2(b1, b2);
            }
}

orc_t *orc_open(const pcamv_params_t *p)
{
    if (p->i_width % 16 || p->i_height % 16) return NULL;
    orc_t *o = calloc(1, sizeof(*o));
    o->p = *p;
    o->mb_w = p->i_width / 16; o->mb_h = p->i_height / 16; o->n_mb = o->mb_w * o->mb_h;
    o->stride = ALIGN16(p->i_width + 2 * PAD); o->lines = p->i_height + 2 * PAD;
    o->cstride = ALIGN16(p->i_width / 2 + 2 * CPAD); o->clines = p->i_height / 2 + 2 * CPAD;
    size_t lsz = (size_t)o->stride * o->lines;
    o->lbuf = calloc(4, lsz);
    for (int i = 0; i < 4; i++) o->luma[i] = o->lbuf + i * lsz + (size_t)o->stride * PAD + PAD;
    for (int i = 0; i < 2; i++) {
        o->cbuf[i] = calloc(1, (size_t)o->cstride * o->clines);
        o->chroma[i] = o->cbuf[i] + (size_t)o->cstride * CPAD + CPAD;
    }
    o->ibuf = calloc(2 * lsz, sizeof(uint16_t));
    o->integral = o->ibuf + (size_t)o->stride * PAD + PAD;
    size_t ysz = (size_t)p->i_width * p->i_height;
    for (int i = 0; i < 3; i++) { o->fenc[i] = malloc(i ? ysz / 4 : ysz); o->frec[i] = malloc(i ? ysz / 4 : ysz); }
    o->mb_type = malloc(o->n_mb);
    o->mv = calloc((size_t)o->n_mb * 16, sizeof(*o->mv));
    o->ref8 = malloc((size_t)o->n_mb * 4);
    o->mvr = calloc(o->n_mb, sizeof(*o->mvr));
    o->prev_mv = calloc((size_t)o->n_mb * 16, sizeof(*o->prev_mv));
    o->prev_ref = malloc((size_t)o->n_mb * 4);
    o->scratch = malloc(sizeof(int16_t) * (4 * 2048 + 64) * 8);
    o->nnz = calloc(o->n_mb, sizeof(*o->nnz));
    o->cbp = calloc(o->n_mb, sizeof(*o->cbp));
    o->mvd = calloc((size_t)o->n_mb * 16, sizeof(*o->mvd));
    build_quant_tables(o);
    orc_srand(&o->rnd, 1);
    return o;
}

void orc_close(orc_t *o)
{
    if (!o) return;
    free(o->lbuf); free(o->cbuf[0]); free(o->cbuf[1]); free(o->ibuf);
    for (int i = 0; i < 3; i++) { free(o->fenc[i]); free(o->frec[i]); }
    free(o->mb_type); free(o->mv); free(o->ref8); free(o->mvr); free(o->prev_mv); free(o->prev_ref); free(o->scratch);
    free(o->nnz); free(o->cbp); free(o->mvd);
    for (int q = 0; q < 52; q++) {
        if (o->cost_mv[q]) free(o->cost_mv[q] - 2 * 4 * 2048);
        for (int j = 0; j < 4; j++) if (o->cost_mv_fpel[q][j]) free(o->cost_mv_fpel[q][j] - 2 * 2048);
    }
    free(o);
}

static const int16_t *get_cost_mv(orc_t *o, int qp)
{
    if (!o->cost_mv[qp]) {
        int16_t *b = malloc((4 * 4 * 2048 + 1) * sizeof(int16_t));
        orc_cost_mv_table(qp, b);
        o->cost_mv[qp] = b + 2 * 4 * 2048;
    }
    /* analyse.c:219-228 */
    if (o->p.i_me_method >= PCAMV_ME_ESA && !o->cost_mv_fpel[qp][0])
        for (int j = 0; j < 4; j++) {
            uint16_t *b = malloc((4 * 2048 + 1) * sizeof(uint16_t));
            o->cost_mv_fpel[qp][j] = b + 2 * 2048;
            for (int i = -2 * 2048; i < 2 * 2048; i++) o->cost_mv_fpel[qp][j][i] = o->cost_mv[qp][i * 4 + j];
        }
    return o->cost_mv[qp];
}

void orc_set_fenc(orc_t *o, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    size_t ysz = (size_t)o->p.i_width * o->p.i_height;
    memcpy(o->fenc[0], y, ysz); memcpy(o->fenc[1], u, ysz / 4); memcpy(o->fenc[2], v, ysz / 4);
}

/* common/frame.c:224-244 with all four bands */
static void expand_border(uint8_t *pix, int stride, int w, int h, int padh, int padv)
{
    for (int y = 0; y < h; y++) {
        memset(pix - padh + (size_t)y * stride, pix[(size_t)y * stride], padh);
        memset(pix + w + (size_t)y * stride, pix[w - 1 + (size_t)y * stride], padh);
    }
    for (int y = 0; y < padv; y++) {
        memcpy(pix - padh - (size_t)(y + 1) * stride, pix - padh, w + 2 * padh);
        memcpy(pix - padh + (size_t)(h + y) * stride, pix - padh + (size_t)(h - 1) * stride, w + 2 * padh);
    }
}

#define TAP(p, d) ((p)[-2 * (d)] + (p)[3 * (d)] - 5 * ((p)[-(d)] + (p)[2 * (d)]) + 20 * ((p)[0] + (p)[(d)]))

/* Whole-frame form of encoder.c:1038-1047: x264_frame_expand_border (frame.c:246),
 * x264_frame_filter -> hpel_filter (mc.c:453-475, 167-190), x264_frame_expand_border_filtered
 * (frame.c:275-301), integral image (mc.c:477-511, 311-345). */
void orc_set_ref(orc_t *o, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                 const int16_t *prev_mv, const int8_t *prev_ref)
{
    int W = o->p.i_width, H = o->p.i_height, st = o->stride;
    for (int r = 0; r < H; r++) memcpy(o->luma[0] + (size_t)r * st, y + (size_t)r * W, W);
    for (int r = 0; r < H / 2; r++) {
        memcpy(o->chroma[0] + (size_t)r * o->cstride, u + (size_t)r * W / 2, W / 2);
        memcpy(o->chroma[1] + (size_t)r * o->cstride, v + (size_t)r * W / 2, W / 2);
    }
    expand_border(o->luma[0], st, W, H, PAD, PAD);
    expand_border(o->chroma[0], o->cstride, W / 2, H / 2, CPAD, CPAD);
    expand_border(o->chroma[1], o->cstride, W / 2, H / 2, CPAD, CPAD);

    /* hpel_filter over rows [-8, H+8), columns [-8, W+8) */
    int16_t *buf = malloc((W + 16 + 8) * sizeof(int16_t));
    for (int r = -8; r < H + 8; r++) {
        const uint8_t *src = o->luma[0] + (ptrdiff_t)r * st - 8;
        uint8_t *dh = o->luma[1] + (ptrdiff_t)r * st - 8, *dv = o->luma[2] + (ptrdiff_t)r * st - 8, *dc = o->luma[3] + (ptrdiff_t)r * st - 8;
        int width = W + 16;
        for (int x = -2; x < width + 3; x++) {
            int t = TAP(src + x, st);
            dv[x] = clip_u8((t + 16) >> 5);
            buf[x + 2] = t;
        }
        for (int x = 0; x < width; x++) dc[x] = clip_u8((TAP(buf + 2 + x, 1) + 512) >> 10);
        for (int x = 0; x < width; x++) dh[x] = clip_u8((TAP(src + x, 1) + 16) >> 5);
    }
    free(buf);
    for (int i = 1; i < 4; i++)
        expand_border(o->luma[i] - 8 * st - 4, st, W + 8, H + 16, PAD - 4, PAD - 8);

    if (o->p.i_me_method >= PCAMV_ME_ESA) {
        int sub8 = !!(o->p.inter & PCAMV_ANALYSE_PSUB8x8);
        uint16_t *integ = o->integral;
        memset(integ - PAD * st - PAD, 0, st * sizeof(uint16_t));
        for (int r = -PAD; r < H + 8 + PAD - 9; r++) {
            const uint8_t *pix = o->luma[0] + (ptrdiff_t)r * st - PAD;
            uint16_t *sum8 = integ + (ptrdiff_t)(r + 1) * st - PAD;
            if (sub8) {
                int vv = pix[0] + pix[1] + pix[2] + pix[3];
                for (int x = 0; x < st - 4; x++) { sum8[x] = vv + sum8[x - st]; vv += pix[x + 4] - pix[x]; }
                sum8 -= 8 * st;
                uint16_t *sum4 = sum8 + (size_t)st * (H + PAD * 2);
                if (r >= 8 - PAD) {
                    for (int x = 0; x < st - 8; x++) sum4[x] = sum8[x + 4 * st] - sum8[x];
                    for (int x = 0; x < st - 8; x++) sum8[x] = sum8[x + 8 * st] + sum8[x + 8 * st + 4] - sum8[x] - sum8[x + 4];
                }
            } else {
                int vv = pix[0] + pix[1] + pix[2] + pix[3] + pix[4] + pix[5] + pix[6] + pix[7];
                for (int x = 0; x < st - 8; x++) { sum8[x] = vv + sum8[x - st]; vv += pix[x + 8] - pix[x]; }
                if (r >= 8 - PAD) {
                    uint16_t *s = sum8 - 8 * st;
                    for (int x = 0; x < st - 8; x++) s[x] = s[x + 8 * st] - s[x];
                }
            }
        }
    }
    o->have_prev = prev_mv != NULL && o->p.i_tscale != 0;
    o->ref_is_inter = prev_mv != NULL;               /* h->fref0[0]->mb_type: the reference is a P picture (the fork codes no intra macroblocks in P slices) */
    if (o->have_prev) {
        memcpy(o->prev_mv, prev_mv, (size_t)o->n_mb * 16 * 2 * sizeof(int16_t));
        memcpy(o->prev_ref, prev_ref, (size_t)o->n_mb * 4);
    }
}
int orc_ref_stride(const orc_t *o) { return o->stride; }
int orc_ref_lines(const orc_t *o) { return o->lines; }
void orc_get_ref_planes(const orc_t *o, uint8_t *out4) { memcpy(out4, o->lbuf, 4 * (size_t)o->stride * o->lines); }
void orc_get_ref_integral(const orc_t *o, uint16_t *out) { memcpy(out, o->ibuf, (size_t)o->stride * o->lines * sizeof(uint16_t)); }

/* ------------------------------------------------------------------------------------------
 * per-macroblock state (the parts of h->mb / h->mb.cache this path touches)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_t *o;
    int mb_x, mb_y, mb_xy;
    int qp, chroma_qp, lambda;
    int mv_min[2], mv_max[2], mv_min_spel[2], mv_max_spel[2], mv_min_fpel[2], mv_max_fpel[2];
    int neighbour;                                   /* bit0 left, 1 top, 2 topright, 3 topleft */
    int type_left, type_top, type_topleft, type_topright;
    int8_t cref[48];
    int16_t cmv[48][2];
    int16_t pskip_mv[2];
    uint8_t fenc[24 * 16], fenc_ih[24 * 16], fdec[27 * 32];
    uint8_t *p_fenc[3], *p_fenc_ih[3], *p_fdec[3];
    uint8_t *fref[6];                                /* 4 luma planes + U,V at this MB's origin */
    uint16_t *integ;
    int i_type, i_partition;
    uint8_t sub_part[4];
    const int16_t *p_cost_mv;
    int b_chroma_me, subme, me_method;
    int b_skip_mc;
    int cbp_luma, cbp_chroma;
    uint8_t nzq[16], nnz[16];                        /* luma 4x4: quantised to non-zero; still non-zero after decimation (h->mb.non_zero_count != 0) */
    /* --subme >= 6 (pcamv_oracle_rd.inc) */
    int mbrd, lambda2, b_fast_intra, fenc_satd_sum, fenc_sa8d_sum, fenc_satd[4][4], fenc_sa8d[2][2];
    int16_t lv[24][16], lvdc[2][4];                  /* h->dct.luma4x4 (zigzag levels; 16..23 chroma with [0] = 0), h->dct.chroma_dc */
    uint8_t nzc[48];                                 /* h->mb.cache.non_zero_count, 0x80 = unavailable */
    int16_t cmvd[48][2];                             /* h->mb.cache.mvd */
    int8_t i4mode[48];                               /* h->mb.cache.intra4x4_pred_mode */
    int cbp_left, cbp_top;                           /* h->mb.cache.i_cbp_left / top, -1 = unavailable */
} mbc_t;
#define NB_LEFT 1
#define NB_TOP 2
#define NB_TOPRIGHT 4
#define NB_TOPLEFT 8

typedef struct {
    int i_pixel, xoff, yoff;
    int i_ref_cost;
    int16_t mvp[2];
    int cost_mv, cost, cost_rec;
    int16_t mv[2];
} me_t;

static inline void cache_mv(mbc_t *m, int x, int y, int w, int h, const int16_t mv[2])
{   /* x264_macroblock_cache_mv_ptr, common/macroblock.h */
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) { m->cmv[SCAN8_0 + x + i + 8 * (y + j)][0] = mv[0]; m->cmv[SCAN8_0 + x + i + 8 * (y + j)][1] = mv[1]; }
}
static inline void cache_ref(mbc_t *m, int x, int y, int w, int h, int ref)
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) m->cref[SCAN8_0 + x + i + 8 * (y + j)] = ref;
}

/* common/macroblock.c:28-101 */
static void predict_mv(mbc_t *m, int idx, int width, int16_t mvp[2])
{
    int i8 = scan8(idx);
    int ref = m->cref[i8];
    int refa = m->cref[i8 - 1], refb = m->cref[i8 - 8], refc = m->cref[i8 - 8 + width];
    const int16_t *a = m->cmv[i8 - 1], *b = m->cmv[i8 - 8], *c = m->cmv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || refc == -2) { refc = m->cref[i8 - 8 - 1]; c = m->cmv[i8 - 8 - 1]; }
    if (m->i_partition == PCAMV_D_16x8) {
        if (idx == 0 && refb == ref) { mvp[0] = b[0]; mvp[1] = b[1]; return; }
        if (idx != 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
    } else if (m->i_partition == PCAMV_D_8x16) {
        if (idx == 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
        if (idx != 0 && refc == ref) { mvp[0] = c[0]; mvp[1] = c[1]; return; }
    }
    int cnt = (refa == ref) + (refb == ref) + (refc == ref);
    if (cnt > 1) { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
    else if (cnt == 1) {
        const int16_t *s = refa == ref ? a : refb == ref ? b : c;
        mvp[0] = s[0]; mvp[1] = s[1];
    } else if (refb == -2 && refc == -2 && refa != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
}
/* common/macroblock.c:103-143 */
static void predict_mv_16x16(mbc_t *m, int ref, int16_t mvp[2])
{
    int refa = m->cref[SCAN8_0 - 1], refb = m->cref[SCAN8_0 - 8], refc = m->cref[SCAN8_0 - 8 + 4];
    const int16_t *a = m->cmv[SCAN8_0 - 1], *b = m->cmv[SCAN8_0 - 8], *c = m->cmv[SCAN8_0 - 8 + 4];
    if (refc == -2) { refc = m->cref[SCAN8_0 - 8 - 1]; c = m->cmv[SCAN8_0 - 8 - 1]; }
    int cnt = (refa == ref) + (refb == ref) + (refc == ref);
    if (cnt > 1) { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
    else if (cnt == 1) {
        const int16_t *s = refa == ref ? a : refb == ref ? b : c;
        mvp[0] = s[0]; mvp[1] = s[1];
    } else if (refb == -2 && refc == -2 && refa != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
}
/* common/macroblock.c:146-163 */
static void predict_mv_pskip(mbc_t *m, int16_t mv[2])
{
    int refa = m->cref[SCAN8_0 - 1], refb = m->cref[SCAN8_0 - 8];
    const int16_t *a = m->cmv[SCAN8_0 - 1], *b = m->cmv[SCAN8_0 - 8];
    if (refa == -2 || refb == -2 || !(refa | a[0] | a[1]) || !(refb | b[0] | b[1])) { mv[0] = mv[1] = 0; }
    else predict_mv_16x16(m, 0, mv);
}
/* note on the test above: the reference ORs the 8-bit ref with the 32-bit packed mv
 * (macroblock.c:154-155); (ref | mvx | mvy) == 0 is the same predicate for int16 mvs. */

/* common/macroblock.c:388-470, P slice, one reference, no lowres */
static int predict_mv_ref16x16(mbc_t *m, int16_t mvc[9][2])
{
    orc_t *o = m->o;
    int i = 0, xy = m->mb_xy, top = xy - o->mb_w;
#define SET(mb) { mvc[i][0] = o->mvr[mb][0]; mvc[i][1] = o->mvr[mb][1]; i++; }
    if ((m->neighbour & NB_LEFT) && o->mb_type[xy - 1] != PCAMV_P_SKIP) SET(xy - 1);
    if (m->neighbour & NB_TOP) {
        if (o->mb_type[top] != PCAMV_P_SKIP) SET(top);
        if ((m->neighbour & NB_TOPLEFT) && o->mb_type[top - 1] != PCAMV_P_SKIP) SET(top - 1);
        if (m->mb_x < o->mb_w - 1 && o->mb_type[top + 1] != PCAMV_P_SKIP) SET(top + 1);
    }
#undef SET
    if (o->have_prev) {
        int scale = o->p.i_tscale;
#define TMVP(dx, dy) { \
        int b4 = 4 * (m->mb_y * 4 * o->mb_w + m->mb_x) + dx * 4 + dy * 4 * (4 * o->mb_w); \
        int b8 = 2 * (m->mb_y * 2 * o->mb_w + m->mb_x) + dx * 2 + dy * 2 * (2 * o->mb_w); \
        if (o->prev_ref[b8] >= 0) { \
            mvc[i][0] = (o->prev_mv[b4][0] * scale + 128) >> 8; \
            mvc[i][1] = (o->prev_mv[b4][1] * scale + 128) >> 8; i++; } }
        TMVP(0, 0);
        if (m->mb_x < o->mb_w - 1) TMVP(1, 0);
        if (m->mb_y < o->mb_h - 1) TMVP(0, 1);
#undef TMVP
    }
    return i;
}

/* x264_macroblock_cache_load (common/macroblock.c:914-1238) + load_pic_pointers (:868-912),
 * restricted to what P analysis reads; x264_mb_analyse_init MV limits (analyse.c:268-318). */
static void mb_load(orc_t *o, mbc_t *m, int mb_x, int mb_y, int qp)
{
    int W = o->p.i_width;
    m->o = o; m->mb_x = mb_x; m->mb_y = mb_y; m->mb_xy = mb_y * o->mb_w + mb_x;
    m->qp = qp; m->lambda = lambda_tab[qp];
    m->chroma_qp = chroma_qp_tab[clip3(qp + o->p.i_chroma_qp_offset, 0, 51)];
    m->subme = o->p.i_subpel_refine; m->me_method = o->p.i_me_method;
    m->b_chroma_me = o->p.b_chroma_me && m->subme >= 5;      /* analyse.c:246-247 */
    m->b_skip_mc = 0;
    m->p_cost_mv = get_cost_mv(o, qp);
    m->p_fenc[0] = m->fenc; m->p_fenc[1] = m->fenc + 16 * 16; m->p_fenc[2] = m->fenc + 16 * 16 + 8;
    m->p_fenc_ih[0] = m->fenc_ih; m->p_fenc_ih[1] = m->fenc_ih + 16 * 16; m->p_fenc_ih[2] = m->fenc_ih + 16 * 16 + 8;
    m->p_fdec[0] = m->fdec + 2 * 32; m->p_fdec[1] = m->fdec + 19 * 32; m->p_fdec[2] = m->fdec + 19 * 32 + 16;
    for (int y = 0; y < 16; y++) memcpy(m->p_fenc[0] + y * 16, o->fenc[0] + (size_t)(mb_y * 16 + y) * W + mb_x * 16, 16);
    for (int c = 1; c < 3; c++)
        for (int y = 0; y < 8; y++) memcpy(m->p_fenc[c] + y * 16, o->fenc[c] + (size_t)(mb_y * 8 + y) * (W / 2) + mb_x * 8, 8);
    memcpy(m->fenc_ih, m->fenc, sizeof(m->fenc));
    for (int k = 0; k < 4; k++) m->fref[k] = o->luma[k] + (size_t)mb_y * 16 * o->stride + mb_x * 16;
    for (int k = 0; k < 2; k++) m->fref[4 + k] = o->chroma[k] + (size_t)mb_y * 8 * o->cstride + mb_x * 8;
    m->integ = o->integral + (size_t)mb_y * 16 * o->stride + mb_x * 16;

    /* neighbours */
    m->neighbour = 0;
    m->type_left = m->type_top = m->type_topleft = m->type_topright = -1;
    int top = m->mb_xy - o->mb_w;
    if (mb_y > 0) { m->neighbour |= NB_TOP; m->type_top = o->mb_type[top]; }
    if (mb_x > 0) { m->neighbour |= NB_LEFT; m->type_left = o->mb_type[m->mb_xy - 1]; }
    if (mb_x < o->mb_w - 1 && mb_y > 0) { m->neighbour |= NB_TOPRIGHT; m->type_topright = o->mb_type[top + 1]; }
    if (mb_x > 0 && mb_y > 0) { m->neighbour |= NB_TOPLEFT; m->type_topleft = o->mb_type[top - 1]; }

    memset(m->cref, -2, sizeof(m->cref));
    memset(m->cmv, 0, sizeof(m->cmv));
    int s4 = 4 * o->mb_w, s8 = 2 * o->mb_w;
    int b4 = 4 * (mb_y * s4 + mb_x), b8 = 2 * (mb_y * s8 + mb_x);
    int t4 = (4 * (mb_y - 1) + 3) * s4 + 4 * mb_x, t8 = (2 * (mb_y - 1) + 1) * s8 + 2 * mb_x;
    if (m->neighbour & NB_TOPLEFT) {
        m->cref[SCAN8_0 - 1 - 8] = o->ref8[t8 - 1];
        m->cmv[SCAN8_0 - 1 - 8][0] = o->mv[t4 - 1][0]; m->cmv[SCAN8_0 - 1 - 8][1] = o->mv[t4 - 1][1];
    }
    if (m->neighbour & NB_TOP)
        for (int i = 0; i < 4; i++) {
            m->cref[SCAN8_0 - 8 + i] = o->ref8[t8 + (i >> 1)];
            m->cmv[SCAN8_0 - 8 + i][0] = o->mv[t4 + i][0]; m->cmv[SCAN8_0 - 8 + i][1] = o->mv[t4 + i][1];
        }
    if (m->neighbour & NB_TOPRIGHT) {
        m->cref[SCAN8_0 + 4 - 8] = o->ref8[t8 + 2];
        m->cmv[SCAN8_0 + 4 - 8][0] = o->mv[t4 + 4][0]; m->cmv[SCAN8_0 + 4 - 8][1] = o->mv[t4 + 4][1];
    }
    if (m->neighbour & NB_LEFT)
        for (int i = 0; i < 4; i++) {
            m->cref[SCAN8_0 - 1 + 8 * i] = o->ref8[b8 - 1 + (i >> 1) * s8];
            m->cmv[SCAN8_0 - 1 + 8 * i][0] = o->mv[b4 - 1 + i * s4][0]; m->cmv[SCAN8_0 - 1 + 8 * i][1] = o->mv[b4 - 1 + i * s4][1];
        }
    predict_mv_pskip(m, m->pskip_mv);

    /* analyse.c:271-317 */
    int fmv = 4 * o->p.i_mv_range;
    m->mv_min[0] = 4 * (-16 * mb_x - 24);
    m->mv_max[0] = 4 * (16 * (o->mb_w - mb_x - 1) + 24);
    m->mv_min_spel[0] = clip3(m->mv_min[0], -fmv, fmv - 1);
    m->mv_max_spel[0] = clip3(m->mv_max[0], -fmv, fmv - 1);
    m->mv_min_fpel[0] = (m->mv_min_spel[0] >> 2) + 5;
    m->mv_max_fpel[0] = (m->mv_max_spel[0] >> 2) - 5;
    m->mv_min[1] = 4 * (-16 * mb_y - 24);
    m->mv_max[1] = 4 * (16 * (o->mb_h - mb_y - 1) + 24);
    m->mv_min_spel[1] = clip3(m->mv_min[1], MAX2(4 * (-512 + 8), -fmv), fmv);
    m->mv_max_spel[1] = clip3(m->mv_max[1], -fmv, fmv - 1);
    m->mv_max_spel[1] = MIN2(m->mv_max_spel[1], fmv * 4);   /* thread_mvy_range == i_fmv_range, 1 thread */
    m->mv_min_fpel[1] = (m->mv_min_spel[1] >> 2) + 5;
    m->mv_max_fpel[1] = (m->mv_max_spel[1] >> 2) - 5;

    m->mbrd = m->subme >= 6;                         /* analyse.c:236 */
    if (!m->mbrd) return;
    m->lambda2 = lambda2_tab[qp];
    /* analyse.c:363-378: no neighbour and no co-located macroblock is ever intra here, so intra is "unlikely" unless the
     * reference picture is an I frame */
    m->b_fast_intra = m->mb_xy > 4 && o->ref_is_inter;
    /* entropy-coder neighbourhood, common/macroblock.c:950-1025, 1170-1200 */
    static const uint8_t bottom[8] = {10, 11, 14, 15, 18, 19, 22, 23}, right[8] = {5, 7, 13, 15, 17, 19, 21, 23};
    static const uint8_t top_pos[8] = {4 + 0 * 8, 5 + 0 * 8, 6 + 0 * 8, 7 + 0 * 8, 1 + 0 * 8, 2 + 0 * 8, 1 + 3 * 8, 2 + 3 * 8};
    static const uint8_t left_pos[8] = {3 + 1 * 8, 3 + 2 * 8, 3 + 3 * 8, 3 + 4 * 8, 0 + 1 * 8, 0 + 2 * 8, 0 + 4 * 8, 0 + 5 * 8};
    /* h->mb.cache is ONE structure for the whole encoder: x264_macroblock_cache_load fills the neighbours' entries and leaves the
     * current macroblock's own non_zero_count / mvd entries as the macroblock coded before it left them.  Nearly everything writes
     * them before reading them; the sub-partition RD trials (x264_rd_cost_part, below) do read them ("the NNZ values used for
     * context selection for future blocks are those left over from previous RDO calls", analyse.c:2158).  mbc_t is one structure
     * for the whole frame too: the inner entries stay, the rest is reset (a frame starts from zeros: what the previous frame's
     * second pass would have left is outside this path). */
    { uint8_t keep_nz[24]; int16_t keep_mvd[16][2];
      for (int i = 0; i < 24; i++) keep_nz[i] = m->nzc[scan8_all[i]];
      for (int i = 0; i < 16; i++) { keep_mvd[i][0] = m->cmvd[scan8(i)][0]; keep_mvd[i][1] = m->cmvd[scan8(i)][1]; }
      memset(m->nzc, 0, sizeof(m->nzc)); memset(m->cmvd, 0, sizeof(m->cmvd)); memset(m->i4mode, -1, sizeof(m->i4mode));
      if (o->p.inter & PCAMV_ANALYSE_PSUB8x8) {
          for (int i = 0; i < 24; i++) m->nzc[scan8_all[i]] = keep_nz[i];
          for (int i = 0; i < 16; i++) { m->cmvd[scan8(i)][0] = keep_mvd[i][0]; m->cmvd[scan8(i)][1] = keep_mvd[i][1]; }
      } }
    for (int i = 0; i < 8; i++) {
        m->nzc[top_pos[i]] = (m->neighbour & NB_TOP) ? o->nnz[top][bottom[i]] : 0x80;
        m->nzc[left_pos[i]] = (m->neighbour & NB_LEFT) ? o->nnz[m->mb_xy - 1][right[i]] : 0x80;
    }
    m->cbp_top = (m->neighbour & NB_TOP) ? o->cbp[top] : -1;
    m->cbp_left = (m->neighbour & NB_LEFT) ? o->cbp[m->mb_xy - 1] : -1;
    for (int i = 0; i < 4; i++) {
        if (m->neighbour & NB_TOP) { m->cmvd[SCAN8_0 - 8 + i][0] = o->mvd[t4 + i][0]; m->cmvd[SCAN8_0 - 8 + i][1] = o->mvd[t4 + i][1]; m->i4mode[SCAN8_0 - 8 + i] = 2; }
        if (m->neighbour & NB_LEFT) { m->cmvd[SCAN8_0 - 1 + 8 * i][0] = o->mvd[b4 - 1 + i * s4][0]; m->cmvd[SCAN8_0 - 1 + 8 * i][1] = o->mvd[b4 - 1 + i * s4][1]; m->i4mode[SCAN8_0 - 1 + 8 * i] = 2; }
    }
    /* intra prediction neighbours: the unfiltered pass-1 reconstruction (x264_fdec_filter_row keeps the line above before
     * the loop filter touches it, encoder.c:1019-1030; the left column is the previous macroblock's fdec buffer) */
    for (int c = 0; c < 3; c++) {
        int w = c ? 8 : 16, pw = c ? W / 2 : W, x0 = mb_x * w, y0 = mb_y * w;
        if (mb_y > 0) for (int x = -1; x < w + w / 2; x++) m->p_fdec[c][x - 32] = o->frec[c][(size_t)(y0 - 1) * pw + clip3(x0 + x, 0, pw - 1)];
        if (mb_x > 0) for (int y = 0; y < w; y++) m->p_fdec[c][y * 32 - 1] = o->frec[c][(size_t)(y0 + y) * pw + x0 - 1];
    }
}

/* ------------------------------------------------------------------------------------------
 * motion search: encoder/me.c
 * ---------------------------------------------------------------------------------------- */
static const int subpel_iterations[][4] = {{0, 0, 0, 0}, {1, 1, 0, 0}, {0, 1, 1, 0}, {0, 2, 1, 0}, {0, 2, 1, 1},
                                           {0, 2, 1, 2}, {0, 0, 2, 2}, {0, 0, 2, 2}, {0, 0, 4, 10}, {0, 0, 4, 10}};
static const int mod6m1[8] = {5, 0, 1, 2, 3, 4, 5, 0};
static const int hex2[8][2] = {{-1, -2}, {-2, 0}, {-1, 2}, {1, 2}, {2, 0}, {1, -2}, {-1, -2}, {-2, 0}};

typedef int (*cmp_fn)(int, const uint8_t *, int, const uint8_t *, int);

typedef struct {
    mbc_t *m; me_t *me;
    const uint8_t *fenc; uint8_t *fref[6]; uint16_t *integ;
    int stride, cstride;
    const int16_t *cmx, *cmy;
    cmp_fn fpelcmp, mbcmp;
    int bw, bh;
} srch_t;

static void srch_init(srch_t *s, mbc_t *m, me_t *me)
{
    orc_t *o = m->o;
    s->m = m; s->me = me; s->stride = o->stride; s->cstride = o->cstride;
    s->fenc = m->p_fenc[0] + me->xoff + me->yoff * 16;
    for (int k = 0; k < 4; k++) s->fref[k] = m->fref[k] + me->xoff + me->yoff * o->stride;
    for (int k = 4; k < 6; k++) s->fref[k] = m->fref[k] + (me->xoff >> 1) + (me->yoff >> 1) * o->cstride;
    s->integ = m->integ + me->xoff + me->yoff * o->stride;
    s->cmx = m->p_cost_mv - me->mvp[0]; s->cmy = m->p_cost_mv - me->mvp[1];
    int satd = m->subme > 1;                                   /* encoder.c:615-625 */
    s->mbcmp = satd ? orc_satd : orc_sad;
    s->fpelcmp = (satd && m->me_method == PCAMV_ME_TESA) ? orc_satd : orc_sad;
    s->bw = pix_w[me->i_pixel]; s->bh = pix_h[me->i_pixel];
}

static inline int cost_fpel(srch_t *s, int mx, int my)
{
    return s->fpelcmp(s->me->i_pixel, s->fenc, 16, s->fref[0] + my * s->stride + mx, s->stride) + s->cmx[mx << 2] + s->cmy[my << 2];
}

/* me.c:689-713 COST_MV_SATD body: returns the cost (with the conditional chroma terms) */
static int cost_qpel_satd(srch_t *s, int mx, int my, int bcost, int b_chroma_me, const uint8_t *fenc_y, const uint8_t *fenc_u, const uint8_t *fenc_v)
{
    uint8_t pix[32 * 18];
    int st = 16, ip = s->me->i_pixel;
    const uint8_t *src = get_ref(pix, &st, s->fref, s->stride, mx, my, s->bw, s->bh);
    int cost = s->mbcmp(ip, fenc_y, 16, src, st) + s->cmx[mx] + s->cmy[my];
    if (b_chroma_me && cost < bcost) {
        orc_mc_chroma(pix, 8, s->fref[4], s->cstride, mx, my, s->bw / 2, s->bh / 2);
        cost += s->mbcmp(ip + 3, fenc_u, 16, pix, 8);
        if (cost < bcost) {
            orc_mc_chroma(pix, 8, s->fref[5], s->cstride, mx, my, s->bw / 2, s->bh / 2);
            cost += s->mbcmp(ip + 3, fenc_v, 16, pix, 8);
        }
    }
    return cost;
}

/* me.c:715-843 */
static void refine_subpel(srch_t *s, int hpel_iters, int qpel_iters, int b_refine_qpel)
{
    mbc_t *m = s->m; me_t *me = s->me;
    const int ip = me->i_pixel, bw = s->bw, bh = s->bh;
    const int b_chroma_me = m->b_chroma_me && ip <= PIX_8x8;
    const uint8_t *fu = m->p_fenc[1] + (me->xoff >> 1) + (me->yoff >> 1) * 16;
    const uint8_t *fv = m->p_fenc[2] + (me->xoff >> 1) + (me->yoff >> 1) * 16;
    uint8_t pix[2][32 * 18];
    int bmx = me->mv[0], bmy = me->mv[1], bcost = me->cost, odir = -1, bdir;

    if (hpel_iters && m->subme < 3) {
        int mx = clip3(me->mvp[0], m->mv_min_spel[0], m->mv_max_spel[0]);
        int my = clip3(me->mvp[1], m->mv_min_spel[1], m->mv_max_spel[1]);
        if ((mx - bmx) | (my - bmy)) {
            int st = 16;
            const uint8_t *src = get_ref(pix[0], &st, s->fref, s->stride, mx, my, bw, bh);
            int cost = s->fpelcmp(ip, s->fenc, 16, src, st) + s->cmx[mx] + s->cmy[my];
            if (cost < bcost) { bcost = cost; bmx = mx; bmy = my; }
        }
    }
    for (int i = hpel_iters; i > 0; i--) {
        int omx = bmx, omy = bmy, st = 32, c;
        const uint8_t *s0 = get_ref(pix[0], &st, s->fref, s->stride, omx, omy - 2, bw, bh + 1);
        const uint8_t *s2 = get_ref(pix[1], &st, s->fref, s->stride, omx - 2, omy, bw + 4, bh);
        const uint8_t *s1 = s0 + st, *s3 = s2 + 1;
        c = s->fpelcmp(ip, s->fenc, 16, s0, st) + s->cmx[omx] + s->cmy[omy - 2]; if (c < bcost) { bcost = c; bmy = omy - 2; }
        c = s->fpelcmp(ip, s->fenc, 16, s1, st) + s->cmx[omx] + s->cmy[omy + 2]; if (c < bcost) { bcost = c; bmy = omy + 2; }
        c = s->fpelcmp(ip, s->fenc, 16, s2, st) + s->cmx[omx - 2] + s->cmy[omy]; if (c < bcost) { bcost = c; bmx = omx - 2; bmy = omy; }
        c = s->fpelcmp(ip, s->fenc, 16, s3, st) + s->cmx[omx + 2] + s->cmy[omy]; if (c < bcost) { bcost = c; bmx = omx + 2; bmy = omy; }
        if (bmx == omx && bmy == omy) break;
    }
    if (!b_refine_qpel) {
        if (bmy > m->mv_max_spel[1]) bmy = m->mv_max_spel[1];
        bcost = COST_MAX;
        int c = cost_qpel_satd(s, bmx, bmy, bcost, b_chroma_me, s->fenc, fu, fv);
        if (c < bcost) bcost = c;
    }
    bdir = -1;
    for (int i = qpel_iters; i > 0; i--) {
        static const int d[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
        odir = bdir;
        int omx = bmx, omy = bmy;
        for (int k = 0; k < 4; k++) {
            if (!(b_refine_qpel || (k ^ 1) != odir)) continue;
            int mx = omx + d[k][0], my = omy + d[k][1];
            int c = cost_qpel_satd(s, mx, my, bcost, b_chroma_me, s->fenc, fu, fv);
            if (c < bcost) { bcost = c; bmx = mx; bmy = my; bdir = k; }
        }
        if (bmx == omx && bmy == omy) break;
    }
    if (bmy > m->mv_max_spel[1]) {
        bmy = m->mv_max_spel[1];
        bcost = COST_MAX;
        int c = cost_qpel_satd(s, bmx, bmy, bcost, b_chroma_me, s->fenc, fu, fv);
        if (c < bcost) bcost = c;
    }
    me->cost = bcost; me->mv[0] = bmx; me->mv[1] = bmy;
    me->cost_mv = s->cmx[bmx] + s->cmy[bmy];
}

/* pixel.c:515-559 */
static int ads_filter(int i_pixel, const int enc_dc[4], const uint16_t *sums, int delta, const uint16_t *cost_mvx, int16_t *mvs, int width, int thresh)
{
    /* ads4: 16x16 ; ads2: 16x8, 8x16, 8x4, 4x8 ; ads1: 8x8, 4x4  (pixel.c:798-804) */
    int kind = (i_pixel == PIX_16x16) ? 4 : (i_pixel == PIX_8x8 || i_pixel == PIX_4x4) ? 1 : 2;
    int nmv = 0;
    for (int i = 0; i < width; i++) {
        const uint16_t *sp = sums + i;
        int ads;
        if (kind == 4)
            ads = abs(enc_dc[0] - sp[0]) + abs(enc_dc[1] - sp[8]) + abs(enc_dc[2] - sp[delta]) + abs(enc_dc[3] - sp[delta + 8]) + cost_mvx[i];
        else if (kind == 2)
            ads = abs(enc_dc[0] - sp[0]) + abs(enc_dc[1] - sp[delta]) + cost_mvx[i];
        else
            ads = abs(enc_dc[0] - sp[0]) + cost_mvx[i];
        if (ads < thresh) mvs[nmv++] = i;
    }
    return nmv;
}

/* me.c:158-666 */
static void me_search(mbc_t *m, me_t *me, int16_t (*mvc)[2], int i_mvc)
{
    orc_t *o = m->o;
    srch_t S, *s = &S;
    srch_init(s, m, me);
    const int ip = me->i_pixel, bw = s->bw, bh = s->bh;
    int i_me_range = o->p.i_me_range;
    int bmx, bmy, bcost, bpred_mx = 0, bpred_my = 0, bpred_cost = COST_MAX, omx, omy, pmx, pmy;
    uint8_t pix[16 * 16];
    int costs[6], i, j, dir;
    const int mv_x_min = m->mv_min_fpel[0], mv_y_min = m->mv_min_fpel[1], mv_x_max = m->mv_max_fpel[0], mv_y_max = m->mv_max_fpel[1];
#define CHECK_MVRANGE(mx, my) ((mx) >= mv_x_min && (mx) <= mv_x_max && (my) >= mv_y_min && (my) <= mv_y_max)
#define COST_MV(mx, my) { int c_ = cost_fpel(s, mx, my); if (c_ < bcost) { bcost = c_; bmx = mx; bmy = my; } }
#define COST_MV_HPEL(mx, my) { int st_ = 16; const uint8_t *src_ = get_ref(pix, &st_, s->fref, s->stride, mx, my, bw, bh); \
        int c_ = s->fpelcmp(ip, s->fenc, 16, src_, st_) + s->cmx[mx] + s->cmy[my]; \
        if (c_ < bpred_cost) { bpred_cost = c_; bpred_mx = mx; bpred_my = my; } }
#define COST_MV_X4(a0, a1, b0, b1, c0, c1, d0, d1) { \
        int c0_ = cost_fpel(s, omx + (a0), omy + (a1)), c1_ = cost_fpel(s, omx + (b0), omy + (b1)); \
        int c2_ = cost_fpel(s, omx + (c0), omy + (c1)), c3_ = cost_fpel(s, omx + (d0), omy + (d1)); \
        if (c0_ < bcost) { bcost = c0_; bmx = omx + (a0); bmy = omy + (a1); } \
        if (c1_ < bcost) { bcost = c1_; bmx = omx + (b0); bmy = omy + (b1); } \
        if (c2_ < bcost) { bcost = c2_; bmx = omx + (c0); bmy = omy + (c1); } \
        if (c3_ < bcost) { bcost = c3_; bmx = omx + (d0); bmy = omy + (d1); } }
#define COST_MV_X3_DIR(a0, a1, b0, b1, c0, c1, out) { \
        (out)[0] = cost_fpel(s, bmx + (a0), bmy + (a1)); (out)[1] = cost_fpel(s, bmx + (b0), bmy + (b1)); (out)[2] = cost_fpel(s, bmx + (c0), bmy + (c1)); }
#define DIA1_ITER(mx, my) { omx = mx; omy = my; COST_MV_X4(0, -1, 0, 1, -1, 0, 1, 0); }
#define CROSS(start, x_max, y_max) { \
        i = start; \
        if ((x_max) <= MIN2(mv_x_max - omx, omx - mv_x_min)) \
            for (; i < (x_max) - 2; i += 4) COST_MV_X4(i, 0, -i, 0, i + 2, 0, -i - 2, 0); \
        for (; i < (x_max); i += 2) { \
            if (omx + i <= mv_x_max) COST_MV(omx + i, omy); \
            if (omx - i >= mv_x_min) COST_MV(omx - i, omy); } \
        i = start; \
        if ((y_max) <= MIN2(mv_y_max - omy, omy - mv_y_min)) \
            for (; i < (y_max) - 2; i += 4) COST_MV_X4(0, i, 0, -i, 0, i + 2, 0, -i - 2); \
        for (; i < (y_max); i += 2) { \
            if (omy + i <= mv_y_max) COST_MV(omx, omy + i); \
            if (omy - i >= mv_y_min) COST_MV(omx, omy - i); } }

    bmx = clip3(me->mvp[0], mv_x_min * 4, mv_x_max * 4);
    bmy = clip3(me->mvp[1], mv_y_min * 4, mv_y_max * 4);
    pmx = (bmx + 2) >> 2; pmy = (bmy + 2) >> 2;
    bcost = COST_MAX;

    if (m->subme >= 3) {
        int sx = bmx, sy = bmy;
        COST_MV_HPEL(bmx, bmy);
        for (i = 0; i < i_mvc; i++)
            if ((mvc[i][0] | mvc[i][1]) && ((sx - mvc[i][0]) | (sy - mvc[i][1]))) {
                /* me.c:206: non-zero candidate that differs from the (packed) start vector */
                int mx = clip3(mvc[i][0], mv_x_min * 4, mv_x_max * 4);
                int my = clip3(mvc[i][1], mv_y_min * 4, mv_y_max * 4);
                COST_MV_HPEL(mx, my);
            }
        bmx = (bpred_mx + 2) >> 2; bmy = (bpred_my + 2) >> 2;
        COST_MV(bmx, bmy);
    } else {
        COST_MV(pmx, pmy);
        bcost -= s->cmx[pmx << 2] + s->cmy[pmy << 2];
        for (i = 0; i < i_mvc; i++) {
            int mx = (mvc[i][0] + 2) >> 2, my = (mvc[i][1] + 2) >> 2;
            if ((mx | my) && ((mx - bmx) | (my - bmy))) {
                mx = clip3(mx, mv_x_min, mv_x_max); my = clip3(my, mv_y_min, mv_y_max);
                COST_MV(mx, my);
            }
        }
    }
    COST_MV(0, 0);

    switch (m->me_method) {
    case PCAMV_ME_DIA:
        i = 0;
        do {
            DIA1_ITER(bmx, bmy);
            if ((bmx == omx) & (bmy == omy)) break;
            if (!CHECK_MVRANGE(bmx, bmy)) break;
        } while (++i < i_me_range);
        break;
    case PCAMV_ME_HEX:
    me_hex2:
        dir = -2;
        COST_MV_X3_DIR(-2, 0, -1, 2, 1, 2, costs);
        COST_MV_X3_DIR(2, 0, 1, -2, -1, -2, costs + 3);
        for (i = 0; i < 6; i++) if (costs[i] < bcost) { bcost = costs[i]; dir = i; }
        if (dir != -2) {
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (i = 1; i < i_me_range / 2 && CHECK_MVRANGE(bmx, bmy); i++) {
                const int odir = mod6m1[dir + 1];
                COST_MV_X3_DIR(hex2[odir + 0][0], hex2[odir + 0][1], hex2[odir + 1][0], hex2[odir + 1][1], hex2[odir + 2][0], hex2[odir + 2][1], costs);
                dir = -2;
                if (costs[0] < bcost) { bcost = costs[0]; dir = odir - 1; }
                if (costs[1] < bcost) { bcost = costs[1]; dir = odir; }
                if (costs[2] < bcost) { bcost = costs[2]; dir = odir + 1; }
                if (dir == -2) break;
                bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            }
        }
        omx = bmx; omy = bmy;
        COST_MV_X4(0, -1, 0, 1, -1, 0, 1, 0);
        COST_MV_X4(-1, -1, -1, 1, 1, -1, 1, 1);
        break;
    case PCAMV_ME_UMH: {
        static const int size_shift[7] = {0, 1, 1, 2, 3, 3, 4};
        int ucost1, ucost2, cross_start = 1;
#define SAD_THRESH(v) (bcost < ((v) >> size_shift[ip]))
        ucost1 = bcost;
        DIA1_ITER(pmx, pmy);
        if (pmx | pmy) DIA1_ITER(0, 0);
        if (ip == PIX_4x4) goto me_hex2;
        ucost2 = bcost;
        if ((bmx | bmy) && ((bmx - pmx) | (bmy - pmy))) DIA1_ITER(bmx, bmy);
        if (bcost == ucost2) cross_start = 3;
        omx = bmx; omy = bmy;
        if (bcost == ucost2 && SAD_THRESH(2000)) {
            COST_MV_X4(0, -2, -1, -1, 1, -1, -2, 0);
            COST_MV_X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500)) break;
            if (bcost == ucost2) {
                int range = (i_me_range >> 1) | 1;
                CROSS(3, range, range);
                COST_MV_X4(-1, -2, 1, -2, -2, -1, 2, -1);
                COST_MV_X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2) break;
                cross_start = range + 2;
            }
        }
        if (i_mvc) {
            static const int range_mul[4][4] = {{3, 3, 4, 4}, {3, 4, 4, 4}, {4, 4, 4, 5}, {4, 4, 5, 6}};
            int mvd, sad_ctx, mvd_ctx, denom = 1;
            if (i_mvc == 1) {
                if (ip == PIX_16x16) mvd = 25;
                else mvd = abs(me->mvp[0] - mvc[0][0]) + abs(me->mvp[1] - mvc[0][1]);
            } else {
                denom = i_mvc - 1; mvd = 0;
                if (ip != PIX_16x16) { mvd = abs(me->mvp[0] - mvc[0][0]) + abs(me->mvp[1] - mvc[0][1]); denom++; }
                for (i = 0; i < i_mvc - 1; i++) mvd += abs(mvc[i][0] - mvc[i + 1][0]) + abs(mvc[i][1] - mvc[i + 1][1]);
            }
            sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            i_me_range = i_me_range * range_mul[mvd_ctx][sad_ctx] / 4;
        }
        CROSS(cross_start, i_me_range, i_me_range / 2);
        COST_MV_X4(-2, -2, -2, 2, 2, -2, 2, 2);
        omx = bmx; omy = bmy;
        i = 1;
        do {
            static const int hex4[16][2] = {{-4, 2}, {-4, 1}, {-4, 0}, {-4, -1}, {-4, -2}, {4, -2}, {4, -1}, {4, 0},
                                            {4, 1},  {4, 2},  {2, 3},  {0, 4},   {-2, 3},  {-2, -3}, {0, -4}, {2, -3}};
            if (4 * i > MIN2(MIN2(mv_x_max - omx, omx - mv_x_min), MIN2(mv_y_max - omy, omy - mv_y_min))) {
                for (j = 0; j < 16; j++) {
                    int mx = omx + hex4[j][0] * i, my = omy + hex4[j][1] * i;
                    if (CHECK_MVRANGE(mx, my)) COST_MV(mx, my);
                }
            } else {
                COST_MV_X4(-4 * i, 2 * i, -4 * i, 1 * i, -4 * i, 0 * i, -4 * i, -1 * i);
                COST_MV_X4(-4 * i, -2 * i, 4 * i, -2 * i, 4 * i, -1 * i, 4 * i, 0 * i);
                COST_MV_X4(4 * i, 1 * i, 4 * i, 2 * i, 2 * i, 3 * i, 0 * i, 4 * i);
                COST_MV_X4(-2 * i, 3 * i, -2 * i, -3 * i, 0 * i, -4 * i, 2 * i, -3 * i);
            }
        } while (++i <= i_me_range / 4);
        if (bmy <= mv_y_max) goto me_hex2;
        break;
    }
    case PCAMV_ME_ESA:
    case PCAMV_ME_TESA: {
        const int min_x = MAX2(bmx - i_me_range, mv_x_min), min_y = MAX2(bmy - i_me_range, mv_y_min);
        const int max_x = MIN2(bmx + i_me_range, mv_x_max), max_y = MIN2(bmy + i_me_range, mv_y_max);
        const int width = (max_x - min_x + 3) & ~3;
        const int stride = s->stride;
        uint16_t *sums_base = s->integ;
        static uint8_t zero[8 * 16];
        int enc_dc[4];
        int sad_size = ip <= PIX_8x8 ? PIX_8x8 : PIX_4x4;
        int delta = pix_w[sad_size];
        int16_t *xs = o->scratch;
        int xn, my;
        /* x264_cost_mv_fpel[qp][-mvp&3] + (-mvp>>2), me.c:513 */
        const uint16_t *cost_fpel_mvx = o->cost_mv_fpel[m->qp][-me->mvp[0] & 3] + (-me->mvp[0] >> 2);
        enc_dc[0] = orc_sad(sad_size, zero, 16, s->fenc, 16);
        enc_dc[1] = orc_sad(sad_size, zero, 16, s->fenc + delta, 16);
        enc_dc[2] = orc_sad(sad_size, zero, 16, s->fenc + delta * 16, 16);
        enc_dc[3] = orc_sad(sad_size, zero, 16, s->fenc + delta + delta * 16, 16);
        if (delta == 4) sums_base += (size_t)stride * (o->p.i_height + PAD * 2);
        if (ip == PIX_16x16 || ip == PIX_8x16 || ip == PIX_4x8) delta *= stride;
        if (ip == PIX_8x16 || ip == PIX_4x8) enc_dc[1] = enc_dc[2];

        if (m->me_method == PCAMV_ME_TESA) {
            typedef struct { int sad; int16_t mx, my; } mvsad_t;
            mvsad_t *mvsads = (mvsad_t *)(xs + ((width + 15) & ~15));
            int nmvsad = 0, limit, sad_thresh = i_me_range <= 16 ? 10 : i_me_range <= 24 ? 11 : 12;
            int bsad = orc_sad(ip, s->fenc, 16, s->fref[0] + bmy * stride + bmx, stride) + s->cmx[bmx << 2] + s->cmy[bmy << 2];
            for (my = min_y; my <= max_y; my++) {
                int ycost = s->cmy[my << 2];
                if (bsad <= ycost) continue;
                bsad -= ycost;
                xn = ads_filter(ip, enc_dc, sums_base + min_x + my * stride, delta, cost_fpel_mvx + min_x, xs, width, bsad * 17 / 16);
                for (i = 0; i < xn; i++) {
                    int mx = min_x + xs[i];
                    int sad = orc_sad(ip, s->fenc, 16, s->fref[0] + mx + my * stride, stride) + cost_fpel_mvx[xs[i]];
                    if (sad < bsad * sad_thresh >> 3) {
                        if (sad < bsad) bsad = sad;
                        mvsads[nmvsad].sad = sad + ycost; mvsads[nmvsad].mx = mx; mvsads[nmvsad].my = my; nmvsad++;
                    }
                }
                bsad += ycost;
            }
            limit = i_me_range / 2;
            if (nmvsad > limit * 2) {
                bsad = bsad * (sad_thresh + 8) >> 4;
                for (i = 0; i < nmvsad && mvsads[i].sad <= bsad; i++);
                for (j = i; j < nmvsad; j++) if (mvsads[j].sad <= bsad) mvsads[i++] = mvsads[j];
                nmvsad = i;
            }
            if (nmvsad > limit) {
                for (i = 0; i < limit; i++) {
                    int bj = i, bs = mvsads[bj].sad;
                    for (j = i + 1; j < nmvsad; j++) if (mvsads[j].sad < bs) { bs = mvsads[j].sad; bj = j; }
                    if (bj > i) { mvsad_t t = mvsads[i]; mvsads[i] = mvsads[bj]; mvsads[bj] = t; }
                }
                nmvsad = limit;
            }
            for (i = 0; i < nmvsad; i++) COST_MV(mvsads[i].mx, mvsads[i].my);
        } else {
            for (my = min_y; my <= max_y; my++) {
                int ycost = s->cmy[my << 2];
                if (bcost <= ycost) continue;
                bcost -= ycost;
                xn = ads_filter(ip, enc_dc, sums_base + min_x + my * stride, delta, cost_fpel_mvx + min_x, xs, width, bcost);
                /* COST_MV_X3_ABS (me.c:107-120): x cost only, y cost re-added afterwards */
                for (i = 0; i < xn - 2; i += 3)
                    for (j = 0; j < 3; j++) {
                        int mx = min_x + xs[i + j];
                        int c = s->fpelcmp(ip, s->fenc, 16, s->fref[0] + mx + my * stride, stride) + s->cmx[mx << 2];
                        if (c < bcost) { bcost = c; bmx = mx; bmy = my; }
                    }
                bcost += ycost;
                for (; i < xn; i++) COST_MV(min_x + xs[i], my);
            }
        }
        break;
    }
    }

    if (bpred_cost < bcost) { me->mv[0] = bpred_mx; me->mv[1] = bpred_my; me->cost = bpred_cost; }
    else { me->mv[0] = bmx << 2; me->mv[1] = bmy << 2; me->cost = bcost; }
    me->cost_mv = s->cmx[me->mv[0]] + s->cmy[me->mv[1]];
    if (bmx == pmx && bmy == pmy && m->subme < 3) me->cost += me->cost_mv;
    if (m->subme >= 2)
        refine_subpel(s, subpel_iterations[m->subme][2], subpel_iterations[m->subme][3], 0);
    else if (me->mv[1] > m->mv_max_spel[1])
        me->mv[1] = m->mv_max_spel[1];
}

/* me.c:669-678 */
static void me_refine_qpel(mbc_t *m, me_t *me)
{
    srch_t S;
    srch_init(&S, m, me);
    if (me->i_pixel <= PIX_8x8) me->cost -= me->i_ref_cost;
    refine_subpel(&S, subpel_iterations[m->subme][0], subpel_iterations[m->subme][1], 1);
}

/* ------------------------------------------------------------------------------------------
 * residual coding: common/dct.c:122-232, common/quant.c:33-109,211-248, encoder/macroblock.c
 * ---------------------------------------------------------------------------------------- */
static void sub4x4_dct(int16_t dct[16], const uint8_t *p1, const uint8_t *p2)   /* p1 stride 16, p2 stride 32 */
{
    int16_t d[4][4], t[4][4];
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y][x] = p1[y * 16 + x] - p2[y * 32 + x];
    for (int i = 0; i < 4; i++) {
        int s03 = d[i][0] + d[i][3], s12 = d[i][1] + d[i][2], d03 = d[i][0] - d[i][3], d12 = d[i][1] - d[i][2];
        t[0][i] = s03 + s12; t[1][i] = 2 * d03 + d12; t[2][i] = s03 - s12; t[3][i] = d03 - 2 * d12;
    }
    for (int i = 0; i < 4; i++) {
        int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
        dct[i * 4 + 0] = s03 + s12; dct[i * 4 + 1] = 2 * d03 + d12; dct[i * 4 + 2] = s03 - s12; dct[i * 4 + 3] = d03 - 2 * d12;
    }
}
static void add4x4_idct(uint8_t *dst, const int16_t dct[16])   /* dst stride 32 */
{
    int16_t d[4][4], t[4][4];
    for (int i = 0; i < 4; i++) {
        int s02 = dct[0 * 4 + i] + dct[2 * 4 + i], d02 = dct[0 * 4 + i] - dct[2 * 4 + i];
        int s13 = dct[1 * 4 + i] + (dct[3 * 4 + i] >> 1), d13 = (dct[1 * 4 + i] >> 1) - dct[3 * 4 + i];
        t[i][0] = s02 + s13; t[i][1] = d02 + d13; t[i][2] = d02 - d13; t[i][3] = s02 - s13;
    }
    for (int i = 0; i < 4; i++) {
        int s02 = t[0][i] + t[2][i], d02 = t[0][i] - t[2][i];
        int s13 = t[1][i] + (t[3][i] >> 1), d13 = (t[1][i] >> 1) - t[3][i];
        d[0][i] = (s02 + s13 + 32) >> 6; d[1][i] = (d02 + d13 + 32) >> 6; d[2][i] = (d02 - d13 + 32) >> 6; d[3][i] = (s02 - s13 + 32) >> 6;
    }
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * 32 + x] = clip_u8(dst[y * 32 + x] + d[y][x]);
}
static int quant4(int16_t dct[16], const uint16_t mf[16], const uint16_t bias[16])
{
    int nz = 0;
    for (int i = 0; i < 16; i++) {
        if (dct[i] > 0) dct[i] = (bias[i] + dct[i]) * mf[i] >> 16;
        else dct[i] = -((bias[i] - dct[i]) * mf[i] >> 16);
        nz |= dct[i];
    }
    return !!nz;
}
static int quant_dc1(int16_t *c, int mf, int bias)
{
    if (*c > 0) *c = (bias + *c) * mf >> 16; else *c = -((bias - *c) * mf >> 16);
    return *c;
}
static void dequant4(int16_t dct[16], int dq[6][16], int qp)
{
    int mf = qp % 6, qbits = qp / 6 - 4;
    if (qbits >= 0) for (int i = 0; i < 16; i++) dct[i] = (dct[i] * dq[mf][i]) << qbits;
    else { int f = 1 << (-qbits - 1); for (int i = 0; i < 16; i++) dct[i] = (dct[i] * dq[mf][i] + f) >> (-qbits); }
}
/* dct.c:528-532, 551: level[i] = dct[x*4+y] */
static const uint8_t zz4[16] = {0 * 4 + 0, 1 * 4 + 0, 0 * 4 + 1, 0 * 4 + 2, 1 * 4 + 1, 2 * 4 + 0, 3 * 4 + 0, 2 * 4 + 1,
                                1 * 4 + 2, 0 * 4 + 3, 1 * 4 + 3, 2 * 4 + 2, 3 * 4 + 1, 3 * 4 + 2, 2 * 4 + 3, 3 * 4 + 3};
/* quant.c:203-239 */
static int decimate_score(const int16_t *l, int n)
{
    static const uint8_t tab[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int idx = n - 1, score = 0;
    while (idx >= 0 && l[idx] == 0) idx--;
    while (idx >= 0) {
        if ((unsigned)(l[idx--] + 1) > 2) return 9;
        int run = 0;
        while (idx >= 0 && l[idx] == 0) { idx--; run++; }
        score += tab[run];
    }
    return score;
}

/* x264_mb_mc for P types: common/macroblock.c:483-508, 560-690 */
static void mc_part(mbc_t *m, int x, int y, int w, int h)
{
    orc_t *o = m->o;
    int i8 = SCAN8_0 + x + 8 * y;
    int mvx = clip3(m->cmv[i8][0], m->mv_min[0], m->mv_max[0]);
    int mvy = clip3(m->cmv[i8][1], m->mv_min[1], m->mv_max[1]);
    orc_mc_luma(m->p_fdec[0] + 4 * y * 32 + 4 * x, 32, m->fref, o->stride, mvx + 16 * x, mvy + 16 * y, 4 * w, 4 * h);
    orc_mc_chroma(m->p_fdec[1] + 2 * y * 32 + 2 * x, 32, m->fref[4] + 2 * y * o->cstride + 2 * x, o->cstride, mvx, mvy, 2 * w, 2 * h);
    orc_mc_chroma(m->p_fdec[2] + 2 * y * 32 + 2 * x, 32, m->fref[5] + 2 * y * o->cstride + 2 * x, o->cstride, mvx, mvy, 2 * w, 2 * h);
}
static void mb_mc(mbc_t *m)
{
    if (m->i_type == PCAMV_P_L0) {
        if (m->i_partition == PCAMV_D_16x16) mc_part(m, 0, 0, 4, 4);
        else if (m->i_partition == PCAMV_D_16x8) { mc_part(m, 0, 0, 4, 2); mc_part(m, 0, 2, 4, 2); }
        else if (m->i_partition == PCAMV_D_8x16) { mc_part(m, 0, 0, 2, 4); mc_part(m, 2, 0, 2, 4); }
    } else if (m->i_type == PCAMV_P_8x8) {
        for (int i = 0; i < 4; i++) {
            int x = 2 * (i & 1), y = 2 * (i >> 1);
            switch (m->sub_part[i]) {
            case PCAMV_D_L0_8x8: mc_part(m, x, y, 2, 2); break;
            case PCAMV_D_L0_8x4: mc_part(m, x, y, 2, 1); mc_part(m, x, y + 1, 2, 1); break;
            case PCAMV_D_L0_4x8: mc_part(m, x, y, 1, 2); mc_part(m, x + 1, y, 1, 2); break;
            case PCAMV_D_L0_4x4: mc_part(m, x, y, 1, 1); mc_part(m, x + 1, y, 1, 1); mc_part(m, x, y + 1, 1, 1); mc_part(m, x + 1, y + 1, 1, 1); break;
            }
        }
    }
}

/* x264_mb_encode_8x8_chroma, inter (encoder/macroblock.c:277-372) */
static void encode_chroma(mbc_t *m)
{
    orc_t *o = m->o;
    int qp = m->chroma_qp, b_decimate = o->p.b_dct_decimate;
    int any_ac = 0;
    for (int ch = 0; ch < 2; ch++) {
        const uint8_t *src = m->p_fenc[1 + ch]; uint8_t *dst = m->p_fdec[1 + ch];
        int16_t dct[4][16], dc[4], lvl[16];
        int score = 0, nz_ac = 0, nz_dc;
        for (int i = 0; i < 4; i++) sub4x4_dct(dct[i], src + (i & 1) * 4 + (i >> 1) * 4 * 16, dst + (i & 1) * 4 + (i >> 1) * 4 * 32);
        /* dct2x2dc (encoder/macroblock.c:71-85): d[0][0],d[0][1],d[1][0],d[1][1] */
        { int d0 = dct[0][0] + dct[1][0], d1 = dct[2][0] + dct[3][0], d2 = dct[0][0] - dct[1][0], d3 = dct[2][0] - dct[3][0];
          dc[0] = d0 + d1; dc[2] = d2 + d3; dc[1] = d0 - d1; dc[3] = d2 - d3;
          dct[0][0] = dct[1][0] = dct[2][0] = dct[3][0] = 0; }
        for (int i = 0; i < 4; i++) {
            int nz = quant4(dct[i], o->quant_mf[1][qp], o->quant_bias[1][qp]);
            m->nzc[scan8_all[16 + i + ch * 4]] = nz;
            if (nz) {
                nz_ac = 1;
                for (int k = 0; k < 16; k++) m->lv[16 + i + ch * 4][k] = lvl[k] = dct[i][zz4[k]];
                dequant4(dct[i], o->dequant_mf, qp);
                if (b_decimate) score += decimate_score(lvl + 1, 15);
            }
        }
        { int mf = o->quant_mf[1][qp][0] >> 1, bias = o->quant_bias[1][qp][0] << 1, nz = 0;
          for (int k = 0; k < 4; k++) nz |= quant_dc1(&dc[k], mf, bias);
          nz_dc = !!nz; }
        m->nzc[scan8_all[25 + ch]] = nz_dc;
        if (nz_dc) { m->lvdc[ch][0] = dc[0]; m->lvdc[ch][1] = dc[2]; m->lvdc[ch][2] = dc[1]; m->lvdc[ch][3] = dc[3]; }   /* zigzag_scan_2x2_dc, :31-38 */
        /* IDCT_DEQUANT_START (encoder/macroblock.c:40-51); dc[] is d[0][0],d[0][1],d[1][0],d[1][1] */
        int d0 = dc[0] + dc[1], d1 = dc[2] + dc[3], d2 = dc[0] - dc[1], d3 = dc[2] - dc[3];
        int dmf = o->dequant_mf[qp % 6][0], qbits = qp / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        if ((b_decimate && score < 7) || !nz_ac) {
            for (int i = 0; i < 4; i++) m->nzc[scan8_all[16 + i + ch * 4]] = 0;
            if (!nz_dc) continue;
            int16_t r[4] = {(int16_t)((d0 + d1) * dmf >> -qbits), (int16_t)((d0 - d1) * dmf >> -qbits),
                            (int16_t)((d2 + d3) * dmf >> -qbits), (int16_t)((d2 - d3) * dmf >> -qbits)};
            /* add8x8_idct_dc (dct.c): each 4x4 gets (dc+32)>>6 added; order dct2x2[0][0],[0][1],[1][0],[1][1] */
            for (int i = 0; i < 4; i++) {
                int v = (r[i] + 32) >> 6;
                uint8_t *p = dst + (i & 1) * 4 + (i >> 1) * 4 * 32;
                for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) p[y * 32 + x] = clip_u8(p[y * 32 + x] + v);
            }
        } else {
            any_ac = 1;
            if (nz_dc) {
                dct[0][0] = (d0 + d1) * dmf >> -qbits; dct[1][0] = (d0 - d1) * dmf >> -qbits;
                dct[2][0] = (d2 + d3) * dmf >> -qbits; dct[3][0] = (d2 - d3) * dmf >> -qbits;
            }
            for (int i = 0; i < 4; i++) add4x4_idct(dst + (i & 1) * 4 + (i >> 1) * 4 * 32, dct[i]);
        }
    }
    m->cbp_chroma = any_ac ? 2 : (m->nzc[scan8_all[25]] | m->nzc[scan8_all[26]]) ? 1 : 0;   /* :364-372 */
}

/* x264_macroblock_encode, inter 4x4-transform branch (encoder/macroblock.c:605-612,690-754,771) and
 * the P_SKIP branch (:511-516, 387-411) */
static void mb_encode(mbc_t *m)
{
    orc_t *o = m->o;
    if (m->i_type == PCAMV_P_SKIP) {
        if (!m->b_skip_mc) {
            int mvx = clip3(m->cmv[SCAN8_0][0], m->mv_min[0], m->mv_max[0]);
            int mvy = clip3(m->cmv[SCAN8_0][1], m->mv_min[1], m->mv_max[1]);
            orc_mc_luma(m->p_fdec[0], 32, m->fref, o->stride, mvx, mvy, 16, 16);
            orc_mc_chroma(m->p_fdec[1], 32, m->fref[4], o->cstride, mvx, mvy, 8, 8);
            orc_mc_chroma(m->p_fdec[2], 32, m->fref[5], o->cstride, mvx, mvy, 8, 8);
        }
        m->cbp_luma = m->cbp_chroma = 0;
        memset(m->nnz, 0, 16);
        for (int i = 0; i < 27; i++) m->nzc[scan8_all[i]] = 0;
        return;
    }
    if (!m->b_skip_mc) mb_mc(m);
    int b_decimate = o->p.b_dct_decimate, qp = m->qp, decimate_mb = 0;
    int16_t dct[16][16], lvl[16];
    m->cbp_luma = 0;
    for (int idx = 0; idx < 16; idx++)
        sub4x4_dct(dct[idx], m->p_fenc[0] + blk_x[idx] * 4 + blk_y[idx] * 4 * 16, m->p_fdec[0] + blk_x[idx] * 4 + blk_y[idx] * 4 * 32);
    for (int i8 = 0; i8 < 4; i8++) {
        int dec8 = 0, cbp = 0;
        for (int i4 = 0; i4 < 4; i4++) {
            int idx = i8 * 4 + i4;
            m->nzq[idx] = 0;
            if (quant4(dct[idx], o->quant_mf[0][qp], o->quant_bias[0][qp])) {
                m->nzq[idx] = 1;
                for (int k = 0; k < 16; k++) m->lv[idx][k] = lvl[k] = dct[idx][zz4[k]];
                dequant4(dct[idx], o->dequant_mf, qp);
                if (b_decimate && dec8 < 6) dec8 += decimate_score(lvl, 16);
                cbp = 1;
            }
        }
        decimate_mb += dec8;
        if (b_decimate) { if (dec8 >= 4) m->cbp_luma |= 1 << i8; }
        else if (cbp) {
            for (int i4 = 0; i4 < 4; i4++) { int idx = i8 * 4 + i4; add4x4_idct(m->p_fdec[0] + blk_x[idx] * 4 + blk_y[idx] * 4 * 32, dct[idx]); }
            m->cbp_luma |= 1 << i8;
        }
    }
    if (b_decimate) {
        if (decimate_mb < 6) m->cbp_luma = 0;
        else
            for (int i8 = 0; i8 < 4; i8++)
                if (m->cbp_luma & (1 << i8))
                    for (int i4 = 0; i4 < 4; i4++) { int idx = i8 * 4 + i4; add4x4_idct(m->p_fdec[0] + blk_x[idx] * 4 + blk_y[idx] * 4 * 32, dct[idx]); }
    }
    for (int idx = 0; idx < 16; idx++) m->nnz[idx] = m->nzq[idx] && ((m->cbp_luma >> (idx >> 2)) & 1);   /* dropped 8x8s / macroblocks are zeroed (macroblock.c:716-751) */
    for (int idx = 0; idx < 16; idx++) m->nzc[scan8_all[idx]] = m->nnz[idx];
    m->nzc[scan8_all[24]] = 0;
    encode_chroma(m);
}
/* NOTE (macroblock.c:725-729): with decimation an 8x8 whose blocks quantise to non-zero but
 * score < 4 is dropped even though its coefficients were dequantised: no IDCT is added. A
 * block that quantises to all-zero contributes nothing either way (IDCT of zeros). */

/* x264_macroblock_probe_skip(h,0): encoder/macroblock.c:809-895 */
static int probe_pskip(mbc_t *m)
{
    orc_t *o = m->o;
    int16_t dct[16], lvl[16];
    int qp = m->qp, decimate = 0;
    int mvx = clip3(m->pskip_mv[0], m->mv_min[0], m->mv_max[0]);
    int mvy = clip3(m->pskip_mv[1], m->mv_min[1], m->mv_max[1]);
    orc_mc_luma(m->p_fdec[0], 32, m->fref, o->stride, mvx, mvy, 16, 16);
    for (int i8 = 0; i8 < 4; i8++)
        for (int i4 = 0; i4 < 4; i4++) {
            int x = (i8 & 1) * 8 + (i4 & 1) * 4, y = (i8 >> 1) * 8 + (i4 >> 1) * 4;
            sub4x4_dct(dct, m->p_fenc[0] + x + y * 16, m->p_fdec[0] + x + y * 32);
            if (!quant4(dct, o->quant_mf[0][qp], o->quant_bias[0][qp])) continue;
            for (int k = 0; k < 16; k++) lvl[k] = dct[zz4[k]];
            decimate += decimate_score(lvl, 16);
            if (decimate >= 6) return 0;
        }
    qp = m->chroma_qp;
    int thresh = (lambda2_tab[qp] + 32) >> 6;
    for (int ch = 0; ch < 2; ch++) {
        const uint8_t *src = m->p_fenc[1 + ch]; uint8_t *dst = m->p_fdec[1 + ch];
        int16_t d4[4][16], dc[4];
        orc_mc_chroma(dst, 32, m->fref[4 + ch], o->cstride, mvx, mvy, 8, 8);
        if (orc_ssd(PIX_8x8, dst, 32, src, 16) < thresh) continue;
        for (int i = 0; i < 4; i++) sub4x4_dct(d4[i], src + (i & 1) * 4 + (i >> 1) * 4 * 16, dst + (i & 1) * 4 + (i >> 1) * 4 * 32);
        { int d0 = d4[0][0] + d4[1][0], d1 = d4[2][0] + d4[3][0], d2 = d4[0][0] - d4[1][0], d3 = d4[2][0] - d4[3][0];
          dc[0] = d0 + d1; dc[2] = d2 + d3; dc[1] = d0 - d1; dc[3] = d2 - d3;
          d4[0][0] = d4[1][0] = d4[2][0] = d4[3][0] = 0; }
        { int mf = o->quant_mf[1][qp][0] >> 1, bias = o->quant_bias[1][qp][0] << 1, nz = 0;
          for (int k = 0; k < 4; k++) nz |= quant_dc1(&dc[k], mf, bias);
          if (nz) return 0; }
        decimate = 0;
        for (int i = 0; i < 4; i++) {
            if (!quant4(d4[i], o->quant_mf[1][qp], o->quant_bias[1][qp])) continue;
            for (int k = 0; k < 16; k++) lvl[k] = d4[i][zz4[k]];
            decimate += decimate_score(lvl + 1, 15);
            if (decimate >= 7) return 0;
        }
    }
    m->b_skip_mc = 1;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * partition analysis: encoder/analyse.c
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    me_t me16x16, me8x8[4], me4x4[4][4], me8x4[4][2], me4x8[4][2], me16x8[2], me8x16[2];
    int16_t mvc[5][2];
    int cost8x8, cost16x8, cost8x16, cost4x4[4], cost8x4[4], cost4x8[4];
    int rd16x16;
} ana_t;

static void me_setup(me_t *me, int i_pixel, int xoff, int yoff) { memset(me, 0, sizeof(*me)); me->i_pixel = i_pixel; me->xoff = xoff; me->yoff = yoff; }

static void update_cache(mbc_t *m, ana_t *a);   /* analyse.c:3703 */
static void cache_fenc_satd(mbc_t *m);
static int rd_cost_mb(mbc_t *m);

/* analyse.c:1122-1204 (one reference).  Returns 1 when the early P_SKIP fired. */
static int analyse_p16x16(mbc_t *m, ana_t *a, int b_try_pskip)
{
    orc_t *o = m->o;
    me_t me; int16_t mvc[9][2]; int i_mvc;
    me_setup(&me, PIX_16x16, 0, 0);
    me.i_ref_cost = 0;                           /* REF_COST(0,0) with one active reference = lambda*bs_size_te(0,..) = 0 */
    predict_mv_16x16(m, 0, me.mvp);
    i_mvc = predict_mv_ref16x16(m, mvc);
    me_search(m, &me, mvc, i_mvc);
    if (b_try_pskip && me.cost - me.cost_mv < 300 * m->lambda &&
        abs(me.mv[0] - m->pskip_mv[0]) + abs(me.mv[1] - m->pskip_mv[1]) <= 1 && probe_pskip(m)) {
        m->i_type = PCAMV_P_SKIP;
        update_cache(m, a);
        return 1;
    }
    me.cost += me.i_ref_cost;
    a->me16x16 = me;
    a->mvc[0][0] = me.mv[0]; a->mvc[0][1] = me.mv[1];
    o->mvr[m->mb_xy][0] = me.mv[0]; o->mvr[m->mb_xy][1] = me.mv[1];
    cache_ref(m, 0, 0, 4, 4, 0);
    m->i_type = PCAMV_P_L0;
    if (m->mbrd) {                                /* analyse.c:1194-1203 */
        cache_fenc_satd(m);
        if (me.mv[0] == m->pskip_mv[0] && me.mv[1] == m->pskip_mv[1]) {
            m->i_partition = PCAMV_D_16x16;
            cache_mv(m, 0, 0, 4, 4, me.mv);
            a->rd16x16 = rd_cost_mb(m);
        }
    }
    return 0;
}
/* analyse.c:1371-1426 */
static void analyse_p8x8(mbc_t *m, ana_t *a)
{
    int i_mvc = 1;
    m->i_partition = PCAMV_D_8x8;
    a->mvc[0][0] = a->me16x16.mv[0]; a->mvc[0][1] = a->me16x16.mv[1];
    for (int i = 0; i < 4; i++) {
        me_t *me = &a->me8x8[i];
        int x8 = i % 2, y8 = i / 2;
        me_setup(me, PIX_8x8, 8 * x8, 8 * y8);
        me->i_ref_cost = 0;
        predict_mv(m, 4 * i, 2, me->mvp);
        me_search(m, me, a->mvc, i_mvc);
        cache_mv(m, 2 * x8, 2 * y8, 2, 2, me->mv);
        a->mvc[i_mvc][0] = me->mv[0]; a->mvc[i_mvc][1] = me->mv[1]; i_mvc++;
        me->cost += me->i_ref_cost;
        me->cost += m->lambda * 1;              /* i_sub_mb_p_cost_table[D_L0_8x8] = 1, analyse.c:179-181 */
    }
    a->cost8x8 = a->me8x8[0].cost + a->me8x8[1].cost + a->me8x8[2].cost + a->me8x8[3].cost;
    /* analyse.c:1422-1423: with cabac one ref cost (0 here) is subtracted */
    for (int i = 0; i < 4; i++) m->sub_part[i] = PCAMV_D_L0_8x8;
}
/* analyse.c:1428-1480 */
static void analyse_p16x8(mbc_t *m, ana_t *a)
{
    m->i_partition = PCAMV_D_16x8;
    for (int i = 0; i < 2; i++) {
        me_t me; int16_t mvc[3][2];
        me_setup(&me, PIX_16x8, 0, 8 * i);
        mvc[0][0] = a->mvc[0][0]; mvc[0][1] = a->mvc[0][1];
        mvc[1][0] = a->mvc[2 * i + 1][0]; mvc[1][1] = a->mvc[2 * i + 1][1];
        mvc[2][0] = a->mvc[2 * i + 2][0]; mvc[2][1] = a->mvc[2 * i + 2][1];
        cache_ref(m, 0, 2 * i, 4, 2, 0);
        predict_mv(m, 8 * i, 4, me.mvp);
        me_search(m, &me, mvc, 3);
        a->me16x8[i] = me;
        cache_mv(m, 0, 2 * i, 4, 2, me.mv);
        cache_ref(m, 0, 2 * i, 4, 2, 0);
    }
    a->cost16x8 = a->me16x8[0].cost + a->me16x8[1].cost;
}
/* analyse.c:1482-1533 */
static void analyse_p8x16(mbc_t *m, ana_t *a)
{
    m->i_partition = PCAMV_D_8x16;
    for (int i = 0; i < 2; i++) {
        me_t me; int16_t mvc[3][2];
        me_setup(&me, PIX_8x16, 8 * i, 0);
        mvc[0][0] = a->mvc[0][0]; mvc[0][1] = a->mvc[0][1];
        mvc[1][0] = a->mvc[i + 1][0]; mvc[1][1] = a->mvc[i + 1][1];
        mvc[2][0] = a->mvc[i + 3][0]; mvc[2][1] = a->mvc[i + 3][1];
        cache_ref(m, 2 * i, 0, 2, 4, 0);
        predict_mv(m, 4 * i, 2, me.mvp);
        me_search(m, &me, mvc, 3);
        a->me8x16[i] = me;
        cache_mv(m, 2 * i, 0, 2, 4, me.mv);
        cache_ref(m, 2 * i, 0, 2, 4, 0);
    }
    a->cost8x16 = a->me8x16[0].cost + a->me8x16[1].cost;
}
/* analyse.c:1535-1567 */
static int sub8x8_chroma_cost(mbc_t *m, ana_t *a, int i8, int pixel)
{
    orc_t *o = m->o;
    uint8_t pix1[16 * 8], *pix2 = pix1 + 8;
    int cs = o->cstride;
    int orr = 4 * (i8 & 1) + 2 * (i8 & 2) * cs, oe = 4 * (i8 & 1) + 2 * (i8 & 2) * 16;
#define CMC(w, h, me, x, y) \
    orc_mc_chroma(&pix1[x + y * 16], 16, m->fref[4] + orr + x + y * cs, cs, (me).mv[0], (me).mv[1], w, h); \
    orc_mc_chroma(&pix2[x + y * 16], 16, m->fref[5] + orr + x + y * cs, cs, (me).mv[0], (me).mv[1], w, h);
    if (pixel == PIX_4x4) { CMC(2, 2, a->me4x4[i8][0], 0, 0); CMC(2, 2, a->me4x4[i8][1], 2, 0); CMC(2, 2, a->me4x4[i8][2], 0, 2); CMC(2, 2, a->me4x4[i8][3], 2, 2); }
    else if (pixel == PIX_8x4) { CMC(4, 2, a->me8x4[i8][0], 0, 0); CMC(4, 2, a->me8x4[i8][1], 0, 2); }
    else { CMC(2, 4, a->me4x8[i8][0], 0, 0); CMC(2, 4, a->me4x8[i8][1], 2, 0); }
#undef CMC
    cmp_fn mbcmp = m->subme > 1 ? orc_satd : orc_sad;
    return mbcmp(PIX_4x4, m->p_fenc[1] + oe, 16, pix1, 16) + mbcmp(PIX_4x4, m->p_fenc[2] + oe, 16, pix2, 16);
}
/* analyse.c:1569-1693 */
static void analyse_sub8x8(mbc_t *m, ana_t *a, int i8, int pixel)
{
    m->i_partition = PCAMV_D_8x8;
    int n = pixel == PIX_4x4 ? 4 : 2, cost = 0;
    for (int k = 0; k < n; k++) {
        int idx = 4 * i8 + (pixel == PIX_8x4 ? 2 * k : k);
        me_t *me = pixel == PIX_4x4 ? &a->me4x4[i8][k] : pixel == PIX_8x4 ? &a->me8x4[i8][k] : &a->me4x8[i8][k];
        me_setup(me, pixel, 4 * blk_x[idx], 4 * blk_y[idx]);
        predict_mv(m, idx, pixel == PIX_8x4 ? 2 : 1, me->mvp);
        int16_t mvc[1][2];
        const me_t *cand = pixel == PIX_4x4 ? &a->me8x8[i8] : &a->me4x4[i8][0];
        mvc[0][0] = cand->mv[0]; mvc[0][1] = cand->mv[1];
        me_search(m, me, mvc, k == 0);
        cache_mv(m, blk_x[idx], blk_y[idx], pixel == PIX_8x4 ? 2 : 1, pixel == PIX_4x8 ? 2 : 1, me->mv);
        cost += me->cost;
    }
    static const int subcost[3] = {5, 3, 3};     /* i_sub_mb_p_cost_table[D_L0_4x4,8x4,4x8], analyse.c:179 */
    int t = pixel == PIX_4x4 ? 0 : pixel == PIX_8x4 ? 1 : 2;
    cost += 0 /* REF_COST */ + m->lambda * subcost[t];
    if (m->b_chroma_me) cost += sub8x8_chroma_cost(m, a, i8, pixel);
    if (pixel == PIX_4x4) a->cost4x4[i8] = cost; else if (pixel == PIX_8x4) a->cost8x4[i8] = cost; else a->cost4x8[i8] = cost;
}
/* analyse.c:1821-1849 */
static void cache_mv_p8x8(mbc_t *m, ana_t *a, int i)
{
    int x = 2 * (i % 2), y = 2 * (i / 2);
    switch (m->sub_part[i]) {
    case PCAMV_D_L0_8x8: cache_mv(m, x, y, 2, 2, a->me8x8[i].mv); break;
    case PCAMV_D_L0_8x4: cache_mv(m, x, y, 2, 1, a->me8x4[i][0].mv); cache_mv(m, x, y + 1, 2, 1, a->me8x4[i][1].mv); break;
    case PCAMV_D_L0_4x8: cache_mv(m, x, y, 1, 2, a->me4x8[i][0].mv); cache_mv(m, x + 1, y, 1, 2, a->me4x8[i][1].mv); break;
    case PCAMV_D_L0_4x4:
        cache_mv(m, x, y, 1, 1, a->me4x4[i][0].mv); cache_mv(m, x + 1, y, 1, 1, a->me4x4[i][1].mv);
        cache_mv(m, x, y + 1, 1, 1, a->me4x4[i][2].mv); cache_mv(m, x + 1, y + 1, 1, 1, a->me4x4[i][3].mv); break;
    }
}
static void update_cache(mbc_t *m, ana_t *a)
{
    switch (m->i_type) {
    case PCAMV_P_L0:
        if (m->i_partition == PCAMV_D_16x16) { cache_ref(m, 0, 0, 4, 4, 0); cache_mv(m, 0, 0, 4, 4, a->me16x16.mv); }
        else if (m->i_partition == PCAMV_D_16x8) { cache_ref(m, 0, 0, 4, 4, 0); cache_mv(m, 0, 0, 4, 2, a->me16x8[0].mv); cache_mv(m, 0, 2, 4, 2, a->me16x8[1].mv); }
        else if (m->i_partition == PCAMV_D_8x16) { cache_ref(m, 0, 0, 4, 4, 0); cache_mv(m, 0, 0, 2, 4, a->me8x16[0].mv); cache_mv(m, 2, 0, 2, 4, a->me8x16[1].mv); }
        break;
    case PCAMV_P_8x8:
        cache_ref(m, 0, 0, 4, 4, 0);
        for (int i = 0; i < 4; i++) cache_mv_p8x8(m, a, i);
        break;
    case PCAMV_P_SKIP:
        m->i_partition = PCAMV_D_16x16;
        cache_ref(m, 0, 0, 4, 4, 0);
        cache_mv(m, 0, 0, 4, 4, m->pskip_mv);
        break;
    }
}

/* MV_SATD_FDEC_IH, analyse.c:2364-2385: SATD of the reconstruction (kept in fenc_ih) against the
 * reference at (mx,my), unconditional chroma terms when chroma ME is on. */
static int mv_satd_rec(mbc_t *m, me_t *me, int mx, int my)
{
    srch_t S; srch_init(&S, m, me);
    uint8_t pix[32 * 18];
    int st = 16, ip = me->i_pixel, bw = S.bw, bh = S.bh;
    const uint8_t *ry = m->p_fenc_ih[0] + me->xoff + me->yoff * 16;
    const uint8_t *ru = m->p_fenc_ih[1] + (me->xoff >> 1) + (me->yoff >> 1) * 16;
    const uint8_t *rv = m->p_fenc_ih[2] + (me->xoff >> 1) + (me->yoff >> 1) * 16;
    const uint8_t *src = get_ref(pix, &st, S.fref, S.stride, mx, my, bw, bh);
    int cost = S.mbcmp(ip, ry, 16, src, st) + S.cmx[mx] + S.cmy[my];
    if (m->b_chroma_me && ip <= PIX_8x8) {
        orc_mc_chroma(pix, 8, S.fref[4], S.cstride, mx, my, bw / 2, bh / 2);
        cost += S.mbcmp(ip + 3, ru, 16, pix, 8);
        orc_mc_chroma(pix, 8, S.fref[5], S.cstride, mx, my, bw / 2, bh / 2);
        cost += S.mbcmp(ip + 3, rv, 16, pix, 8);
    }
    return cost;
}
static void store_rec_ih(mbc_t *m)   /* analyse.c:3880-3890 */
{
    for (int y = 0; y < 16; y++) memcpy(m->p_fenc_ih[0] + y * 16, m->p_fdec[0] + y * 32, 16);
    for (int c = 1; c < 3; c++) for (int y = 0; y < 8; y++) memcpy(m->p_fenc_ih[c] + y * 16, m->p_fdec[c] + y * 32, 8);
}

/* x264_ih_get_mv_cost, analyse.c:2391-2550; tables analyse.c:2562-2565 */
static const int8_t d_mv[12][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-2, 1}, {-1, 2}, {1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}};
static const int8_t d_nb[9][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}, {0, 0}};
static int rca_mv_cost(mbc_t *m, ana_t *a, me_t *me, int16_t *m_x, int16_t *m_y)
{
    const float beta1 = 1.4, beta2 = 4;
    int16_t bmx = me->mv[0], bmy = me->mv[1];
    int cost = 0, min_cost = COST_MAX, nb_cost[9];
    int b_1_neighbor = 0, b_error_pos = 0;
    update_cache(m, a); mb_encode(m); store_rec_ih(m);
    for (int k = 0; k < 9; k++) {
        cost = mv_satd_rec(m, me, bmx + d_nb[k][0], bmy + d_nb[k][1]);
        nb_cost[k] = cost;
        if (cost < min_cost) min_cost = cost;
    }
    me->cost_rec = nb_cost[8];
    const int want_optimal = !(min_cost < me->cost_rec);
    min_cost = COST_MAX; *m_x = 0; *m_y = 0;
    int ii_best = -1;
    for (int ii = 0; ii < 12; ii++) {
        int min1 = COST_MAX, bx1 = bmx + d_mv[ii][0], by1 = bmy + d_mv[ii][1];
        me->mv[0] = bx1; me->mv[1] = by1;
        update_cache(m, a); mb_encode(m); store_rec_ih(m);
        for (int k = 0; k < 9; k++) {
            cost = mv_satd_rec(m, me, bx1 + d_nb[k][0], by1 + d_nb[k][1]);
            if (cost < min1) min1 = cost;
        }
        int is_opt = (min1 == cost);
        if (is_opt == want_optimal && cost < min_cost) { min_cost = cost; *m_x = d_mv[ii][0]; *m_y = d_mv[ii][1]; ii_best = ii; }
        if (ii == 3 && min_cost != COST_MAX) break;
    }
    if (min_cost == COST_MAX) {
        b_error_pos = 1; b_1_neighbor = 1;
        *m_x = 0; *m_y = 0;
        for (int k = 0; k < 4; k++) if (nb_cost[k] < min_cost) { min_cost = nb_cost[k]; *m_x = d_nb[k][0]; *m_y = d_nb[k][1]; }
    } else b_1_neighbor = ii_best <= 3;
    int cost_opt = min_cost > me->cost_rec ? min_cost - me->cost_rec : 1;
    if (!b_1_neighbor) cost_opt = beta1 * (float)cost_opt;
    else if (b_error_pos) cost_opt = beta2 * (float)cost_opt;
    me->mv[0] = bmx; me->mv[1] = bmy;
    update_cache(m, a);
    return cost_opt;
}

#include "pcamv_oracle_rd.inc"

/* x264_macroblock_analyse, P slice (analyse.c:2613-2868, 3471, 3518-3689) */
static void analyse_mb(mbc_t *m, int embed, pcamv_mb_t *out)
{
    orc_t *o = m->o;
    ana_t A, *a = &A;
    memset(a, 0, sizeof(*a));
    a->rd16x16 = a->cost8x8 = a->cost16x8 = a->cost8x16 = COST_MAX;       /* analyse.c:321-332 */
    for (int i = 0; i < 4; i++) a->cost4x4[i] = a->cost8x4[i] = a->cost4x8[i] = COST_MAX;
    int b_skip = 0, b_try_pskip = 0, i_cost;
    unsigned flags = o->p.inter;
    memset(out, 0, sizeof(*out));
    out->pskip_mv[0] = m->pskip_mv[0]; out->pskip_mv[1] = m->pskip_mv[1];
    out->i_qp = m->qp;
    for (int i = 0; i < 4; i++) m->sub_part[i] = PCAMV_D_L0_8x8;
    m->i_partition = PCAMV_D_16x16;

    if (o->p.b_fast_pskip) {
        if (m->subme >= 3) b_try_pskip = 1;
        else if (m->type_left == PCAMV_P_SKIP || m->type_top == PCAMV_P_SKIP || m->type_topleft == PCAMV_P_SKIP || m->type_topright == PCAMV_P_SKIP)
            b_skip = probe_pskip(m);
    }
    if (b_skip) { m->i_type = PCAMV_P_SKIP; m->i_partition = PCAMV_D_16x16; }
    else if (!analyse_p16x16(m, a, b_try_pskip)) {
        int i_type = PCAMV_P_L0, i_partition = PCAMV_D_16x16;
        if (flags & PCAMV_ANALYSE_PSUB16x16) analyse_p8x8(m, a);
        i_cost = a->me16x16.cost;
        if ((flags & PCAMV_ANALYSE_PSUB16x16) && a->cost8x8 < a->me16x16.cost) {
            if (flags & PCAMV_ANALYSE_PSUB8x8) {
                i_type = PCAMV_P_8x8; i_partition = PCAMV_D_8x8; i_cost = a->cost8x8;
                for (int i = 0; i < 4; i++) {
                    analyse_sub8x8(m, a, i, PIX_4x4);
                    if (a->cost4x4[i] < a->me8x8[i].cost) {
                        int c8 = a->cost4x4[i];
                        m->sub_part[i] = PCAMV_D_L0_4x4;
                        analyse_sub8x8(m, a, i, PIX_8x4);
                        if (a->cost8x4[i] < c8) { c8 = a->cost8x4[i]; m->sub_part[i] = PCAMV_D_L0_8x4; }
                        analyse_sub8x8(m, a, i, PIX_4x8);
                        if (a->cost4x8[i] < c8) { c8 = a->cost4x8[i]; m->sub_part[i] = PCAMV_D_L0_4x8; }
                        i_cost += c8 - a->me8x8[i].cost;
                    }
                    cache_mv_p8x8(m, a, i);
                }
                a->cost8x8 = i_cost;
            }
        }
        if ((flags & PCAMV_ANALYSE_PSUB16x16) &&
            a->cost8x8 < a->me16x16.cost + a->me8x8[1].cost_mv + a->me8x8[2].cost_mv) {
            analyse_p16x8(m, a);
            if (a->cost16x8 < i_cost) { i_cost = a->cost16x8; i_type = PCAMV_P_L0; i_partition = PCAMV_D_16x8; }
            analyse_p8x16(m, a);
            if (a->cost8x16 < i_cost) { i_cost = a->cost8x16; i_type = PCAMV_P_L0; i_partition = PCAMV_D_8x16; }
        }
        m->i_partition = i_partition;
        if (m->mbrd) {
            /* analyse.c:2749-2752, 2809-2850: no quarter-pel refinement; the intra SATD cost (never the intra mode) bounds the
             * RD trials; the partition is decided by x264_rd_cost_mb */
            int i_satd_inter = i_cost, i16, i4;
            if (m->b_chroma_me) {
                int c8 = intra_chroma_cost(m);
                analyse_intra(m, i_cost - c8, &i16, &i4);
                i16 += c8; i4 += c8;
            } else analyse_intra(m, i_cost, &i16, &i4);
            int i_satd_intra = MIN2(i16, i4);
            if (getenv("ORC_DBG_MB") && atoi(getenv("ORC_DBG_MB")) == m->mb_xy)
                fprintf(stderr, "orc intra mb %d inter %d i16 %d i4 %d fast %d costs 16x16 %d 8x8 %d 16x8 %d 8x16 %d\n", m->mb_xy, i_satd_inter, i16, i4, m->b_fast_intra, a->me16x16.cost, a->cost8x8, a->cost16x8, a->cost8x16);
            analyse_p_rd(m, a, MIN2(i_satd_inter, i_satd_intra), embed);
            i_type = PCAMV_P_L0; i_partition = PCAMV_D_16x16; i_cost = a->me16x16.cost;
            if (a->cost16x8 < i_cost) { i_cost = a->cost16x8; i_partition = PCAMV_D_16x8; }
            if (a->cost8x16 < i_cost) { i_cost = a->cost8x16; i_partition = PCAMV_D_8x16; }
            if (embed && a->cost8x8 < i_cost) { i_cost = a->cost8x8; i_partition = PCAMV_D_8x8; i_type = PCAMV_P_8x8; }   /* analyse.c:2841-2842: only while embedding */
            m->i_partition = i_partition;
        } else
        if (i_partition == PCAMV_D_16x16) me_refine_qpel(m, &a->me16x16);
        else if (i_partition == PCAMV_D_16x8) { me_refine_qpel(m, &a->me16x8[0]); me_refine_qpel(m, &a->me16x8[1]); }
        else if (i_partition == PCAMV_D_8x16) { me_refine_qpel(m, &a->me8x16[0]); me_refine_qpel(m, &a->me8x16[1]); }
        else
            for (int i = 0; i < 4; i++)
                switch (m->sub_part[i]) {
                case PCAMV_D_L0_8x8: me_refine_qpel(m, &a->me8x8[i]); break;
                case PCAMV_D_L0_8x4: me_refine_qpel(m, &a->me8x4[i][0]); me_refine_qpel(m, &a->me8x4[i][1]); break;
                case PCAMV_D_L0_4x8: me_refine_qpel(m, &a->me4x8[i][0]); me_refine_qpel(m, &a->me4x8[i][1]); break;
                case PCAMV_D_L0_4x4: for (int k = 0; k < 4; k++) me_refine_qpel(m, &a->me4x4[i][k]); break;
                }
        m->i_type = i_type;
    }
    update_cache(m, a);

    /* pass-1 record, analyse.c:3518-3689 */
    out->i_type = m->i_type; out->i_partition = m->i_partition;
    memcpy(out->i_sub_partition, m->sub_part, 4);
    if (embed && m->i_type != PCAMV_P_SKIP) {
        out->used = 1;
        for (int i = 0; i < 16; i++) { out->mv[i][0] = m->cmv[scan8(i)][0]; out->mv[i][1] = m->cmv[scan8(i)][1]; out->ref[i] = m->cref[scan8(i)]; }
#define RCA(mep, slot) { me_t *me_ = (mep); int16_t bx_ = me_->mv[0], by_ = me_->mv[1], dx_ = 0, dy_ = 0; \
            int c_ = rca_mv_cost(m, a, me_, &dx_, &dy_); \
            out->mv_stego[slot][0] = bx_ + dx_; out->mv_stego[slot][1] = by_ + dy_; out->inter_stego_cost[slot] = c_; }
        if (m->i_type == PCAMV_P_8x8) {
            for (int i = 0; i < 4; i++)
                switch (m->sub_part[i]) {
                case PCAMV_D_L0_8x8: RCA(&a->me8x8[i], i * 4); break;
                case PCAMV_D_L0_4x8: for (int j = 0; j < 2; j++) RCA(&a->me4x8[i][j], i * 4 + j); break;
                case PCAMV_D_L0_8x4: for (int j = 0; j < 2; j++) RCA(&a->me8x4[i][j], i * 4 + 2 * j); break;
                case PCAMV_D_L0_4x4: for (int j = 0; j < 4; j++) RCA(&a->me4x4[i][j], i * 4 + j); break;
                }
        } else {
            if (m->i_partition == PCAMV_D_16x16) RCA(&a->me16x16, 0)
            else if (m->i_partition == PCAMV_D_8x16) { for (int j = 0; j < 2; j++) RCA(&a->me8x16[j], j * 4); }
            else if (m->i_partition == PCAMV_D_16x8) { for (int j = 0; j < 2; j++) RCA(&a->me16x8[j], j * 8); }
        }
#undef RCA
    }
    /* the pass-1 encode of the MB (encoder.c after analyse) and x264_macroblock_cache_save */
    mb_encode(m);
    if (m->mbrd) {
        entropy_commit(m);
        if (o->dbg_state_hash) {
            uint32_t hsh = 2166136261u;
            for (int i = 0; i < 460; i++) hsh = (hsh ^ o->cabac_state[i]) * 16777619u;
            o->dbg_state_hash[m->mb_xy] = hsh;
        }
        if (o->dbg_state && m->mb_xy == o->dbg_mb) memcpy(o->dbg_state, o->cabac_state, 460);
    }
    for (int i = 0; i < 16; i++) {
        if (!embed || m->i_type == PCAMV_P_SKIP) { out->mv[i][0] = m->cmv[scan8(i)][0]; out->mv[i][1] = m->cmv[scan8(i)][1]; out->ref[i] = m->cref[scan8(i)]; }
    }
    if (m->i_type != PCAMV_P_SKIP) { out->mvr16[0] = o->mvr[m->mb_xy][0]; out->mvr16[1] = o->mvr[m->mb_xy][1]; }
    if (m->i_type != PCAMV_P_8x8) memset(out->i_sub_partition, PCAMV_D_L0_8x8, 4);
    o->mb_type[m->mb_xy] = m->i_type;
    int s4 = 4 * o->mb_w, s8 = 2 * o->mb_w, b4 = 4 * (m->mb_y * s4 + m->mb_x), b8 = 2 * (m->mb_y * s8 + m->mb_x);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) { o->mv[b4 + y * s4 + x][0] = m->cmv[SCAN8_0 + x + 8 * y][0]; o->mv[b4 + y * s4 + x][1] = m->cmv[SCAN8_0 + x + 8 * y][1]; }
    o->ref8[b8] = m->cref[scan8(0)]; o->ref8[b8 + 1] = m->cref[scan8(4)]; o->ref8[b8 + s8] = m->cref[scan8(8)]; o->ref8[b8 + s8 + 1] = m->cref[scan8(12)];
    int W = o->p.i_width;
    for (int y = 0; y < 16; y++) memcpy(o->frec[0] + (size_t)(m->mb_y * 16 + y) * W + m->mb_x * 16, m->p_fdec[0] + y * 32, 16);
    for (int c = 1; c < 3; c++)
        for (int y = 0; y < 8; y++) memcpy(o->frec[c] + (size_t)(m->mb_y * 8 + y) * (W / 2) + m->mb_x * 8, m->p_fdec[c] + y * 32, 8);
}

void orc_set_debug(orc_t *o, uint32_t *state_hash, int dump_mb, uint8_t *dump_state) { o->dbg_state_hash = state_hash; o->dbg_mb = dump_mb; o->dbg_state = dump_state; }
int orc_analyse_pframe(orc_t *o, int qp, int embed, pcamv_mb_t *out_mb, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v)
{
    mbc_t *m = malloc(sizeof(*m));
    memset(m, 0, sizeof(*m));
    memset(o->mb_type, PCAMV_P_SKIP, o->n_mb);
    if (o->p.i_subpel_refine >= 6) {
        orc_cabac_init_p(o->cabac_state, qp);                                 /* x264_cabac_context_init at the slice start, encoder.c:1227 */
    }
    for (int my = 0; my < o->mb_h; my++)
        for (int mx = 0; mx < o->mb_w; mx++) {
            mb_load(o, m, mx, my, qp);
            analyse_mb(m, embed, &out_mb[my * o->mb_w + mx]);
        }
    free(m);
    size_t ysz = (size_t)o->p.i_width * o->p.i_height;
    if (rec_y) { memcpy(rec_y, o->frec[0], ysz); memcpy(rec_u, o->frec[1], ysz / 4); memcpy(rec_v, o->frec[2], ysz / 4); }
    return 0;
}

void orc_me_search(orc_t *o, int qp, int mb_x, int mb_y, int i_pixel, int xoff, int yoff,
                   const int16_t mvp[2], const int16_t (*mvc)[2], int i_mvc, int16_t out_mv[2], int out_cost[2])
{
    mbc_t *m = malloc(sizeof(*m));
    me_t me; int16_t lm[16][2];
    mb_load(o, m, mb_x, mb_y, qp);
    me_setup(&me, i_pixel, xoff, yoff);
    me.mvp[0] = mvp[0]; me.mvp[1] = mvp[1];
    for (int i = 0; i < i_mvc; i++) { lm[i][0] = mvc[i][0]; lm[i][1] = mvc[i][1]; }
    me_search(m, &me, lm, i_mvc);
    out_mv[0] = me.mv[0]; out_mv[1] = me.mv[1]; out_cost[0] = me.cost; out_cost[1] = me.cost_mv;
    free(m);
}

/* ------------------------------------------------------------------------------------------
 * embedding stage: encoder.c:1561-1855 (cover, costs, MVC adjustment, message, flips)
 * PARITY UNPINNED at this level: encoder.c cannot be compiled here.
 * ---------------------------------------------------------------------------------------- */
/* carrier order of one MB: slots into mv[] / inter_stego_cost[] (encoder.c:1566-1647) */
static int carrier_slots(const pcamv_mb_t *mb, int slots[16])
{
    int n = 0;
    if (!mb->used) return 0;
    if (mb->i_type == PCAMV_P_8x8) {
        for (int i = 0; i < 4; i++)
            switch (mb->i_sub_partition[i]) {
            case PCAMV_D_L0_8x8: slots[n++] = i * 4; break;
            case PCAMV_D_L0_4x8: slots[n++] = i * 4; slots[n++] = i * 4 + 1; break;
            case PCAMV_D_L0_8x4: slots[n++] = i * 4; slots[n++] = i * 4 + 2; break;
            case PCAMV_D_L0_4x4: for (int j = 0; j < 4; j++) slots[n++] = i * 4 + j; break;
            }
    } else if (mb->i_type == PCAMV_P_L0) {
        if (mb->i_partition == PCAMV_D_16x16) slots[n++] = 0;
        else if (mb->i_partition == PCAMV_D_8x16) { slots[n++] = 0; slots[n++] = 4; }
        else if (mb->i_partition == PCAMV_D_16x8) { slots[n++] = 0; slots[n++] = 8; }
    }
    return n;
}
static inline int is01(int d) { return d == 0 || d == 1; }

int orc_embed_pframe(orc_t *o, const pcamv_mb_t *mbs, float emrate, const uint8_t *message, int message_len, pcamv_embed_t *out)
{
    const float alpha_loc = 1, alpha_com = 0, mvc_c1 = 2, mvc_c2 = 0.7;   /* encoder.c:1651-1653 */
    int n = 0;
    for (int xy = 0; xy < o->n_mb; xy++) {
        int slots[16], k = carrier_slots(&mbs[xy], slots);
        for (int i = 0; i < k; i++) {
            out->cover[n] = (mbs[xy].mv[slots[i]][0] + mbs[xy].mv[slots[i]][1]) & 1;
            out->rho[n] = (float)mbs[xy].inter_stego_cost[slots[i]];
            n++;
        }
    }
    /* MVC adjustment + blend, encoder.c:1650-1819 */
    int len = 0;
    for (int xy = 0; xy < o->n_mb; xy++) {
        const pcamv_mb_t *mb = &mbs[xy];
        if (!mb->used) continue;
#define MVD(a, b, c) abs(mb->mv[a][c] - mb->mv[b][c])
        if (mb->i_type == PCAMV_P_8x8) {
            const uint8_t *sp = mb->i_sub_partition;
            if (sp[0] == PCAMV_D_L0_8x8 && sp[1] == PCAMV_D_L0_8x8 && sp[2] == PCAMV_D_L0_8x8 && sp[3] == PCAMV_D_L0_8x8) {
                int c = is01(MVD(0, 4, 0)) + is01(MVD(4, 12, 0)) + is01(MVD(12, 8, 0)) + is01(MVD(8, 0, 0)) +
                        is01(MVD(0, 4, 1)) + is01(MVD(4, 12, 1)) + is01(MVD(12, 8, 1)) + is01(MVD(8, 0, 1));
                for (int j = 0; j < 4; j++) out->rho[len + j] = out->rho[len + j] * (mvc_c2 * c + 1);
            }
            for (int i = 0; i < 4; i++)
                switch (sp[i]) {
                case PCAMV_D_L0_8x8:
                    out->rho[len] = alpha_loc * out->rho[len] + alpha_com * 0.0f; len++; break;
                case PCAMV_D_L0_4x8:
                case PCAMV_D_L0_8x4: {
                    int b = sp[i] == PCAMV_D_L0_4x8 ? 4 * i + 1 : 4 * i + 2;
                    if (MVD(4 * i, b, 0) + MVD(4 * i, b, 1) < 2) { out->rho[len] *= mvc_c1; out->rho[len + 1] *= mvc_c1; }
                    for (int j = 0; j < 2; j++) { out->rho[len] = alpha_loc * out->rho[len] + alpha_com * 0.0f; len++; }
                    break; }
                case PCAMV_D_L0_4x4: {
                    int q = 4 * i;
                    int c = is01(MVD(q, q + 1, 0)) + is01(MVD(q + 1, q + 3, 0)) + is01(MVD(q + 2, q + 3, 0)) + is01(MVD(q, q + 2, 0)) +
                            is01(MVD(q, q + 1, 1)) + is01(MVD(q + 1, q + 3, 1)) + is01(MVD(q + 2, q + 3, 1)) + is01(MVD(q, q + 2, 1));
                    for (int j = 0; j < 4; j++) out->rho[len + j] = out->rho[len + j] * (mvc_c2 * c + 1);
                    for (int j = 0; j < 4; j++) { out->rho[len] = alpha_loc * out->rho[len] + alpha_com * 0.0f; len++; }
                    break; }
                }
        } else if (mb->i_type == PCAMV_P_L0) {
            if (mb->i_partition == PCAMV_D_16x16) { out->rho[len] = alpha_loc * out->rho[len] + alpha_com * 0.0f; len++; }
            else {
                int b = mb->i_partition == PCAMV_D_8x16 ? 4 : 8;
                if (MVD(0, b, 0) + MVD(0, b, 1) < 2) { out->rho[len] *= mvc_c1; out->rho[len + 1] *= mvc_c1; }
                for (int j = 0; j < 2; j++) { out->rho[len] = alpha_loc * out->rho[len] + alpha_com * 0.0f; len++; }
            }
        }
#undef MVD
    }
    out->n = n;
    int an = emrate > 1 ? (int)emrate : (int)(emrate * n);   /* encoder.c:1828-1836 */
    out->m = an;
    for (int i = 0; i < an; i++)
        out->message[i] = message ? (i < message_len ? message[i] : 0) : (orc_rand(&o->rnd) & 1);
    memset(out->stego, 0, n); memset(out->flip, 0, n);
    out->stc_ok = an > 0 ? orc_stc_embed(out->cover, n, out->message, an, out->rho, out->stego, 10) : 0;
    out->num_flip = 0;
    for (int i = 0; i < n; i++)
        if (out->cover[i] ^ out->stego[i]) { out->flip[i] = 1; out->num_flip++; }
    return 0;
}

/* pass-2 substitution, analyse.c:3001-3107: swap in mv_stego where flip[k] == 1 */
/* the carrier slot that owns 4x4 block i (x264 block order) of a macroblock */
static int carrier_of_block(const pcamv_mb_t *mb, int i)
{
    if (mb->i_type == PCAMV_P_8x8) {
        int i8 = i >> 2, j = i & 3;
        switch (mb->i_sub_partition[i8]) {
        case PCAMV_D_L0_8x8: return 4 * i8;
        case PCAMV_D_L0_4x8: return 4 * i8 + (j & 1);
        case PCAMV_D_L0_8x4: return 4 * i8 + (j & 2);
        default: return i;
        }
    }
    if (mb->i_partition == PCAMV_D_8x16) return blk_x[i] < 2 ? 0 : 4;
    if (mb->i_partition == PCAMV_D_16x8) return blk_y[i] < 2 ? 0 : 8;
    return 0;
}
void orc_final_mvs(const orc_t *o, const pcamv_embed_t *e, pcamv_mb_t *mbs)
{
    int k = 0;
    for (int xy = 0; xy < o->n_mb; xy++) {
        int slots[16], c = carrier_slots(&mbs[xy], slots);
        for (int i = 0; i < c; i++, k++)
            if (e->flip[k] == 1) { mbs[xy].mv[slots[i]][0] = mbs[xy].mv_stego[slots[i]][0]; mbs[xy].mv[slots[i]][1] = mbs[xy].mv_stego[slots[i]][1]; }
    }
}

/* ------------------------------------------------------------------------------------------
 * syndrome-trellis code: embed.h
 * ---------------------------------------------------------------------------------------- */
#include "stc_mats.inc"

typedef struct { long hold; } lcg_t;
/* ------------------------------------------------------------------------------------------
 * Pass 2 and the loop filter.  Semantics: every carrier macroblock keeps its pass-1 type /
 * partition and takes its MVs from the record, mv_stego where the flip map says so (analyse.c:2870-3107); a
 * P_SKIP macroblock takes the skip prediction from its FINAL neighbours (what the reference's pass 2 leaves
 * when its skip probe fires again, and what a decoder derives); x264_macroblock_encode; then
 * x264_frame_deblock_row (common/frame.c:627-798) for inter macroblocks with the 4x4 transform.
 * ---------------------------------------------------------------------------------------- */
/* H.264 Tables 8-16 / 8-17 (alpha, beta, tc0 by indexA / indexB and bS), as common/frame.c:383-423 holds them */
static const uint8_t dbk_alpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
                                      32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const uint8_t dbk_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
                                     9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
static const int8_t dbk_tc0[52][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1}, {0, 1, 1}, {1, 1, 1},
    {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3}, {2, 2, 3}, {2, 2, 4}, {2, 3, 4},
    {2, 3, 4}, {3, 3, 5}, {3, 4, 6}, {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10}, {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16},
    {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

/* one 4-sample group of an edge: pix -> first q sample, xs = step across the edge, ys = step along it */
static void dbk_luma4(uint8_t *pix, int xs, int ys, int alpha, int beta, int tc0)
{
    for (int d = 0; d < 4; d++, pix += ys) {
        int p2 = pix[-3 * xs], p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs];
        if (abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta) {
            int tc = tc0, delta;
            if (abs(p2 - p0) < beta) { pix[-2 * xs] = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0); tc++; }
            if (abs(q2 - q0) < beta) { pix[xs] = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0); tc++; }
            delta = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
            pix[-xs] = clip_u8(p0 + delta); pix[0] = clip_u8(q0 - delta);
        }
    }
}
static void dbk_chroma2(uint8_t *pix, int xs, int ys, int alpha, int beta, int tc)
{
    for (int d = 0; d < 2; d++, pix += ys) {
        int p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
        if (abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta) {
            int delta = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
            pix[-xs] = clip_u8(p0 + delta); pix[0] = clip_u8(q0 - delta);
        }
    }
}
/* deblock_edge (frame.c:594-612): bS 1..3 -> tc0 column bS-1; bS 0 -> tc -1 (luma: skip; chroma: tc0+1 = 0 -> skip) */
static void dbk_edge(uint8_t *pl, uint8_t *pu, uint8_t *pv, int xs, int ys, int cxs, int cys, const uint8_t bS[4], int qp, int qpc, int do_chroma)
{
    int alpha = dbk_alpha[qp], beta = dbk_beta[qp];
    if (alpha && beta)
        for (int i = 0; i < 4; i++)
            if (bS[i]) dbk_luma4(pl + 4 * i * ys, xs, ys, alpha, beta, dbk_tc0[qp][bS[i] - 1]);
    if (do_chroma) {
        alpha = dbk_alpha[qpc]; beta = dbk_beta[qpc];
        if (alpha && beta)
            for (int i = 0; i < 4; i++)
                if (bS[i]) {
                    int tc = dbk_tc0[qpc][bS[i] - 1] + 1;
                    dbk_chroma2(pu + 2 * i * cys, cxs, cys, alpha, beta, tc);
                    dbk_chroma2(pv + 2 * i * cys, cxs, cys, alpha, beta, tc);
                }
    }
}
static void deblock_frame(orc_t *o, uint8_t *py, uint8_t *pu, uint8_t *pv, const uint8_t *nnz, int qp)
{
    const int W = o->p.i_width, CW = W / 2, s4 = 4 * o->mb_w, s8 = 2 * o->mb_w;
    const int qpc = chroma_qp_tab[clip3(qp + o->p.i_chroma_qp_offset, 0, 51)];
    const int qp_thresh = 15 - (o->p.i_chroma_qp_offset > 0 ? o->p.i_chroma_qp_offset : 0);
    for (int my = 0; my < o->mb_h; my++)
        for (int mx = 0; mx < o->mb_w; mx++) {
            const int xy = my * o->mb_w + mx, type = o->mb_type[xy];
            int edge_end = type == PCAMV_P_SKIP ? 1 : 4;
            const int no_sub8x8 = type != PCAMV_P_8x8 || !(o->p.inter & PCAMV_ANALYSE_PSUB8x8);
            if (qp <= qp_thresh) edge_end = 1;
            uint8_t *y0 = py + (size_t)16 * my * W + 16 * mx, *u0 = pu + (size_t)8 * my * CW + 8 * mx, *v0 = pv + (size_t)8 * my * CW + 8 * mx;
            for (int dir = 0; dir < 2; dir++)
                for (int edge = (dir ? my == 0 : mx == 0) ? 1 : 0; edge < edge_end; edge++) {
                    const int nxy = edge ? xy : (dir ? xy - o->mb_w : xy - 1);
                    uint8_t bS[4] = {0, 0, 0, 0};
                    for (int i = 0; i < 4; i++) {
                        int x = dir == 0 ? edge : i, y = dir == 0 ? i : edge;
                        int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                        /* non_zero_count is indexed in x264 block order: block of (x,y) */
                        int bi = (x & 1) + 2 * (y & 1) + 4 * (x >> 1) + 8 * (y >> 1), bn = (xn & 1) + 2 * (yn & 1) + 4 * (xn >> 1) + 8 * (yn >> 1);
                        if (nnz[xy * 16 + bi] || nnz[nxy * 16 + bn]) bS[i] = 2;
                        else if (!(edge & no_sub8x8)) {
                            if ((i & no_sub8x8) && bS[i - 1] != 2) bS[i] = bS[i - 1];
                            else {
                                /* 4x4 / 8x8 positions in the frame's motion field */
                                int gx = 4 * mx + x, gy = 4 * my + y, gxn = dir == 0 ? gx - 1 : gx, gyn = dir == 0 ? gy : gy - 1;
                                const int16_t *a = o->mv[gy * s4 + gx], *b = o->mv[gyn * s4 + gxn];
                                if (o->ref8[(gy >> 1) * s8 + (gx >> 1)] != o->ref8[(gyn >> 1) * s8 + (gxn >> 1)] || abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4) bS[i] = 1;
                            }
                        }
                    }
                    if (!(bS[0] | bS[1] | bS[2] | bS[3])) continue;
                    if (dir == 0) dbk_edge(y0 + 4 * edge, u0 + 2 * edge, v0 + 2 * edge, 1, W, 1, CW, bS, qp, qpc, !(edge & 1));
                    else dbk_edge(y0 + (size_t)4 * edge * W, u0 + (size_t)2 * edge * CW, v0 + (size_t)2 * edge * CW, W, 1, CW, 1, bS, qp, qpc, !(edge & 1));
                }
        }
}

/* mbs: the pass-1 record (with mv_stego); flips: one flag per carrier in embedding order.  out (may alias
 * nothing): final type / partition / MVs per macroblock; nnz: [n_mb][16] flags in x264 block order; rec /
 * dbk: reconstruction before / after the loop filter.  Leaves the final motion field in the context. */
int orc_pass2_pframe(orc_t *o, int qp, const pcamv_mb_t *mbs, const uint8_t *flips, int n_flips, pcamv_mb_t *out, uint8_t *nnz,
                     uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint8_t *dbk_y, uint8_t *dbk_u, uint8_t *dbk_v)
{
    mbc_t *m = malloc(sizeof(*m));
    int k = 0;
    memset(o->mb_type, PCAMV_P_SKIP, o->n_mb);
    for (int my = 0; my < o->mb_h; my++)
        for (int mx = 0; mx < o->mb_w; mx++) {
            const int xy = my * o->mb_w + mx;
            const pcamv_mb_t *r = &mbs[xy];
            pcamv_mb_t *f = &out[xy];
            mb_load(o, m, mx, my, qp);
            *f = *r;
            m->i_type = r->i_type; m->i_partition = r->i_partition;
            memcpy(m->sub_part, r->i_sub_partition, 4);
            if (r->i_type == PCAMV_P_SKIP) {
                m->i_partition = PCAMV_D_16x16;
                for (int i = 0; i < 16; i++) { f->mv[i][0] = m->pskip_mv[0]; f->mv[i][1] = m->pskip_mv[1]; f->ref[i] = 0; }
                f->pskip_mv[0] = m->pskip_mv[0]; f->pskip_mv[1] = m->pskip_mv[1];
            } else {
                int slots[16], n = carrier_slots(r, slots);
                for (int j = 0; j < n; j++, k++) {
                    if (k >= n_flips) { free(m); return -1; }
                    if (!flips[k]) continue;
                    /* the 4x4 blocks the carrier covers: same value in every slot the partition owns */
                    int s0 = slots[j];
                    for (int i = 0; i < 16; i++)
                        if (carrier_of_block(r, i) == s0) { f->mv[i][0] = r->mv_stego[s0][0]; f->mv[i][1] = r->mv_stego[s0][1]; }
                }
            }
            for (int i = 0; i < 16; i++) { m->cmv[scan8(i)][0] = f->mv[i][0]; m->cmv[scan8(i)][1] = f->mv[i][1]; m->cref[scan8(i)] = 0; }
            mb_encode(m);
            memcpy(nnz + (size_t)xy * 16, m->nnz, 16);
            o->mb_type[xy] = m->i_type;
            int s4 = 4 * o->mb_w, s8 = 2 * o->mb_w, b4 = 4 * (my * s4 + mx), b8 = 2 * (my * s8 + mx), W = o->p.i_width;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) { o->mv[b4 + y * s4 + x][0] = m->cmv[SCAN8_0 + x + 8 * y][0]; o->mv[b4 + y * s4 + x][1] = m->cmv[SCAN8_0 + x + 8 * y][1]; }
            o->ref8[b8] = o->ref8[b8 + 1] = o->ref8[b8 + s8] = o->ref8[b8 + s8 + 1] = 0;
            for (int y = 0; y < 16; y++) memcpy(o->frec[0] + (size_t)(my * 16 + y) * W + mx * 16, m->p_fdec[0] + y * 32, 16);
            for (int c = 1; c < 3; c++)
                for (int y = 0; y < 8; y++) memcpy(o->frec[c] + (size_t)(my * 8 + y) * (W / 2) + mx * 8, m->p_fdec[c] + y * 32, 8);
        }
    free(m);
    size_t ysz = (size_t)o->p.i_width * o->p.i_height;
    if (rec_y) { memcpy(rec_y, o->frec[0], ysz); memcpy(rec_u, o->frec[1], ysz / 4); memcpy(rec_v, o->frec[2], ysz / 4); }
    deblock_frame(o, o->frec[0], o->frec[1], o->frec[2], nnz, qp);
    if (dbk_y) { memcpy(dbk_y, o->frec[0], ysz); memcpy(dbk_u, o->frec[1], ysz / 4); memcpy(dbk_v, o->frec[2], ysz / 4); }
    return k;
}

static int stc_myrand(lcg_t *s) { return (int)(((s->hold = s->hold * 214013L + 2531011L) >> 16) & 0x7fff); }   /* embed.h:134-139 */

/* embed.h:276-306.  The reference's LCG state is a process-wide static (embed.h:134) that
 * persists across calls; a fresh state per call is identical for every width in 2..20. */
static int stc_get_matrix(int width, int height, uint32_t *cols, lcg_t *lcg)
{
    if (width >= 2 && width <= 20 && height >= 7 && height <= 12) {
        memcpy(cols, &stc_mats[(height - 7) * 400 + (width - 1) * 20], width * sizeof(uint32_t));
        return 1;
    }
    if ((1 << (height - 2)) < width) return 0;
    uint32_t mask = (1u << (height - 2)) - 1, bop = (1u << (height - 1)) + 1;
    for (int i = 0; i < width; i++) {
        uint32_t r; int j;
        for (j = -1; j < i;) {
            r = ((stc_myrand(lcg) & mask) << 1) + bop;
            for (j = 0; j < i; j++) if (cols[j] == r) break;
        }
        cols[i] = r;
    }
    return 1;
}

typedef struct { int shorter, longer; int *width; uint8_t *which; uint32_t *cols[2]; } stc_sched_t;
/* The reference's LCG (embed.h:134) is one process-wide state that only advances when a width
 * outside 2..20 asks for random columns (payload 1 bit/MV or below 1/20).  g_lcg mirrors it for
 * embedding; the extractor replays from the state the last embed call started with. */
static lcg_t g_lcg = {1}, g_lcg_last = {1};
void orc_stc_lcg_reset(long state) { g_lcg.hold = g_lcg_last.hold = state; }
static int stc_schedule(stc_sched_t *s, int n, int m, int hgt, int extracting)     /* embed.h:340-393 */
{
    lcg_t tmp = g_lcg_last, *plcg = &tmp;
    if (!extracting) { g_lcg_last = g_lcg; plcg = &g_lcg; }
#define lcg (*plcg)
    double invalpha = (double)n / m;
    if (invalpha < 1) return 0;
    s->shorter = (int)floor(invalpha); s->longer = (int)ceil(invalpha);
    s->cols[0] = malloc(MAX2(s->shorter, 1) * sizeof(uint32_t)); s->cols[1] = malloc(MAX2(s->longer, 1) * sizeof(uint32_t));
    if (!stc_get_matrix(s->shorter, hgt, s->cols[0], &lcg) || !stc_get_matrix(s->longer, hgt, s->cols[1], &lcg)) { free(s->cols[0]); free(s->cols[1]); return 0; }
    s->width = malloc(m * sizeof(int)); s->which = malloc(m);
    int worm = 0;
    for (int i = 0; i < m; i++) {
        if (worm + s->longer <= (i + 1) * invalpha + 0.5) { s->which[i] = 1; s->width[i] = s->longer; worm += s->longer; }
        else { s->which[i] = 0; s->width[i] = s->shorter; worm += s->shorter; }
    }
#undef lcg
    return 1;
}
static void stc_sched_free(stc_sched_t *s) { free(s->cols[0]); free(s->cols[1]); free(s->width); free(s->which); }

/* embed.h:309-548 */
int orc_stc_embed(const uint8_t *cover, int n, const uint8_t *msg, int m, const float *rho, uint8_t *stego, int hgt)
{
    if (hgt > 31 || m <= 0) return 0;
    int height = 1 << hgt;
    uint32_t colmask = height - 1;
    height = (height + 31) & ~31;
    int parts = height >> 5;
    stc_sched_t sc;
    if (!stc_schedule(&sc, n, m, hgt, 0)) return 0;
    uint32_t *path = calloc((size_t)n * parts, sizeof(uint32_t));
    uint8_t *path8 = (uint8_t *)path, *done = calloc(height, 1);
    float *prices = malloc(height * sizeof(float));
    float inf; { uint32_t b = 0x7F800000; memcpy(&inf, &b, 4); }
    double total = 0;
    for (int i = 0; i < height; i++) prices[i] = inf;
    prices[0] = 0.0f;
    size_t pi8 = 0; int index = 0;
    for (int i2 = 0; i2 < m; i2++) {
        for (int k = 0; k < sc.width[i2]; k++, index++) {
            uint32_t column = sc.cols[sc.which[i2]][k] & colmask;
            float c1, c2;
            if (cover[index] == 0) { c1 = 0.0f; c2 = rho[index]; } else { c1 = rho[index]; c2 = 0.0f; }
            total += rho[index];
            for (int st = 0; st < height; st++) {
                if (done[st]) continue;
                int alt = st ^ column;
                float v1 = prices[st], v2 = prices[alt], v3 = v1, v4 = v2;
                done[st] = 1; done[alt] = 1;
                v1 = v1 + c1; v2 = v2 + c2; v3 = v3 + c2; v4 = v4 + c1;
                v1 = v1 <= v2 ? v1 : v2;
                v4 = v3 <= v4 ? v3 : v4;
                prices[st] = v1; prices[alt] = v4;
                if (v1 == v2) path8[pi8 + (st >> 3)] ^= 1 << (st & 7);
                if (v4 == v3) path8[pi8 + (alt >> 3)] ^= 1 << (alt & 7);
            }
            memset(done, 0, height);
            pi8 += parts << 2;
        }
        int i = msg[i2] == 0 ? 0 : 1, l;
        for (l = 0; i < height; i += 2, l++) prices[l] = prices[i];
        if (m - i2 <= hgt) colmask >>= 1;
        for (; l < height; l++) prices[l] = inf;
    }
    double totalprice = prices[0];
    free(prices); free(done);
    if (totalprice >= total) { free(path); stc_sched_free(&sc); return 0; }
    /* backward pass, embed.h:509-540 */
    size_t pidx = (size_t)index * parts - parts;    /* one path row per processed column */
    index--;
    uint32_t state = 0; colmask = 0;
    for (int i2 = m - 1; i2 >= 0; i2--)
        for (int k = sc.width[i2] - 1; k >= 0; k--, index--) {
            if (k == sc.width[i2] - 1) {
                state = (state << 1) | msg[i2];
                if (m - i2 <= hgt) colmask = (colmask << 1) | 1;
            }
            if (path[pidx + (state >> 5)] & (1u << (state & 31))) { stego[index] = 1; state ^= sc.cols[sc.which[i2]][k] & colmask; }
            else stego[index] = 0;
            pidx -= parts;
        }
    free(path); stc_sched_free(&sc);
    return 1;
}
/* Extractor (absent from the reference): H*y over GF(2) with the same schedule. */
int orc_stc_extract(const uint8_t *stego, int n, int m, int hgt, uint8_t *msg)
{
    stc_sched_t sc;
    if (m <= 0 || !stc_schedule(&sc, n, m, hgt, 1)) return 0;
    memset(msg, 0, m);
    int index = 0;
    for (int i2 = 0; i2 < m; i2++)
        for (int k = 0; k < sc.width[i2]; k++, index++)
            if (stego[index]) {
                uint32_t col = sc.cols[sc.which[i2]][k];
                for (int b = 0; b < hgt && i2 + b < m; b++) msg[i2 + b] ^= (col >> b) & 1;
            }
    stc_sched_free(&sc);
    return 1;
}

/* glibc random_r TYPE_3 (x^31 + x^3 + 1) as rand() uses it */
void orc_srand(orc_rand_t *s, unsigned seed)
{
    int32_t *r = s->r;
    if (seed == 0) seed = 1;
    r[0] = seed;
    for (int i = 1; i < 31; i++) {
        long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    s->f = 3; s->b = 0;
    for (int i = 0; i < 310; i++) (void)orc_rand(s);
}
int orc_rand(orc_rand_t *s)
{
    uint32_t v = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
    s->r[s->f] = (int32_t)v;
    int res = (v >> 1) & 0x7fffffff;
    if (++s->f >= 31) s->f = 0;
    if (++s->b >= 31) s->b = 0;
    return res;
}
