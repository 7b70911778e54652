/*
 * pcamv_host_tables.h -- host-side construction of the per-QP tables the kernels read
 * (part of the product library; also compiled into the test-only emulation driver).
 *
 *   lambda / lambda2        encoder/analyse.c:148-167
 *   MV bit-cost table       encoder/analyse.c:193-209 (p_cost_mv, generated with the same libm
 *                           expression, uploaded once per QP)
 *   quant / dequant scales  common/set.c:26-43, 68-174 for the flat (default) matrices
 *   chroma QP mapping       H.264 table 8-15 (encoder.c:735)
 */
#ifndef PCAMV_HOST_TABLES_H
#define PCAMV_HOST_TABLES_H
#include <math.h>
#include <stdint.h>
#include "pcamv_common.h"
#include "pcamv_entropy_tables.h"

static const int pcamv_lambda_tab[52] = {
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
    6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
static const int pcamv_lambda2_tab[52] = {
    14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322,
    2925, 3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628,
    117964, 148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
static const uint8_t pcamv_chroma_qp_tab[52] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};


/* lambda * (2*log2(i+1) + 0.718 + !!i) + .5 evaluated exactly as the reference's expression:
 * (float)log(x) / log(2.0) in double, times 2, plus float 0.718f, truncated to int16 */
static inline void pcamv_build_cost_mv(int qp, int16_t *out)
{
    const int lambda = pcamv_lambda_tab[qp];
    int16_t *c = out + PCAMV_COST_MV_CENTRE;
    for (int i = 0; i <= PCAMV_COST_MV_CENTRE; i++) {
        double bits = ((float)log((double)(i + 1))) / (log((double)2)) * 2 + 0.718f + !!i;
        c[-i] = c[i] = (int16_t)(lambda * bits + .5f);
    }
}

static inline int pcamv_quant_class_scale(int q6, int cls)
{
    static const int qnt[6][3] = {{13107, 8066, 5243}, {11916, 7490, 4660}, {10082, 6554, 4194},
                                  {9362, 5825, 3647},  {8192, 5243, 3355},  {7282, 4559, 2893}};
    return qnt[q6][cls];
}
static inline int pcamv_dequant_class_scale(int q6, int cls)
{
    static const int deq[6][3] = {{10, 13, 16}, {11, 14, 18}, {13, 16, 20}, {14, 18, 23}, {16, 20, 25}, {18, 23, 29}};
    return deq[q6][cls] * 16;
}

/* fill the QP-dependent part of a FrameDev */
static inline void pcamv_frame_set_qp(FrameDev *F, const pcamv_params_t *p, int qp)
{
    F->qp = qp;
    int cq = qp + p->i_chroma_qp_offset; cq = cq < 0 ? 0 : cq > 51 ? 51 : cq;
    F->chroma_qp = pcamv_chroma_qp_tab[cq];
    F->lambda = pcamv_lambda_tab[qp];
    F->lambda2_chroma = pcamv_lambda2_tab[F->chroma_qp];
    const int dz[2] = {32 - p->i_luma_deadzone[0], 32 - 21};
    const int qps[2] = {qp, F->chroma_qp};
    for (int cat = 0; cat < 2; cat++)
        for (int cls = 0; cls < 3; cls++) {
            int base = pcamv_quant_class_scale(qps[cat] % 6, cls), s = qps[cat] / 6 - 1, j;
            j = s < 0 ? base << -s : s == 0 ? base : (base + (1 << (s - 1))) >> s;
            int b1 = ((dz[cat] << 10) + (j >> 1)) / j, b2 = (1 << 15) / j;
            F->q_mf[cat][cls] = j; F->q_bias[cat][cls] = b1 < b2 ? b1 : b2;
        }
    for (int cls = 0; cls < 3; cls++) {
        F->dq_mf[cls] = pcamv_dequant_class_scale(qp % 6, cls);
        F->dq_mf_c[cls] = pcamv_dequant_class_scale(F->chroma_qp % 6, cls);
    }
    /* --subme >= 6: lambda2 of the RD cost, the intra luma quantiser (CQM_4IY, common/set.c:77) of the 4x4 intra analysis */
    F->lambda2 = pcamv_lambda2_tab[qp];
    for (int cls = 0; cls < 3; cls++) {
        int base = pcamv_quant_class_scale(qp % 6, cls), s = qp / 6 - 1, j;
        j = s < 0 ? base << -s : s == 0 ? base : (base + (1 << (s - 1))) >> s;
        int b1 = (((32 - p->i_luma_deadzone[1]) << 10) + (j >> 1)) / j, b2 = (1 << 15) / j;
        F->q_mf_i[cls] = j; F->q_bias_i[cls] = b1 < b2 ? b1 : b2;
    }
}
/* CABAC context states at the start of a P slice (common/cabac.c:787-805, cabac_init_idc 0) and the per-(state, bin)
 * table the size-only coder walks: x264's 8.8 fixed-point entropy << 8 | next state (common/cabac.c:718-781) */
static inline void pcamv_build_cabac_init(int qp, uint8_t *out /* [464] */)
{
    for (int i = 0; i < 460; i++) {
        int v = ((pcamv_cabac_init_p[2 * i] * qp) >> 4) + pcamv_cabac_init_p[2 * i + 1];
        out[i] = (uint8_t)(v < 1 ? 1 : v > 126 ? 126 : v);
    }
    out[460] = out[461] = out[462] = out[463] = 0;
}
static inline void pcamv_build_cabac_tab(uint32_t *out /* [256] */)
{
    for (int i = 0; i < 256; i++) out[i] = (uint32_t)pcamv_cabac_entropy[i] << 8 | pcamv_cabac_transition[i];
}
static inline void pcamv_frame_set_params(FrameDev *F, const pcamv_params_t *p)
{
    F->w = p->i_width; F->h = p->i_height; F->mb_w = F->w / 16; F->mb_h = F->h / 16; F->n_mb = F->mb_w * F->mb_h;
    F->stride = (F->w + 2 * PCAMV_PAD + 15) & ~15; F->lines = F->h + 2 * PCAMV_PAD;
    F->cstride = (F->w / 2 + 2 * PCAMV_CPAD + 15) & ~15; F->clines = F->h / 2 + 2 * PCAMV_CPAD;
    F->plane_size = (long long)PCAMV_LSTRIPS(F->stride) * PCAMV_LROW * F->lines;   /* strip layout, pcamv_common.h */
    F->lskip = PCAMV_LROW * F->lines - PCAMV_LSW;
    F->me_method = p->i_me_method; F->me_range = p->i_me_range; F->subme = p->i_subpel_refine; F->mv_range = p->i_mv_range;
    F->b_chroma_me = p->b_chroma_me && p->i_subpel_refine >= 5;     /* analyse.c:246-247 */
    F->b_fast_pskip = p->b_fast_pskip; F->b_dct_decimate = p->b_dct_decimate; F->b_cabac = p->b_cabac;
    F->inter = p->inter; F->tscale = p->i_tscale; F->chroma_qp_offset = p->i_chroma_qp_offset;
    F->b_mbrd = p->i_subpel_refine >= 6;                              /* analyse.c:236 */
    F->psy_rd = F->b_mbrd ? p->i_psy_rd : 0;                          /* encoder.c:513-515 */
}
#endif
