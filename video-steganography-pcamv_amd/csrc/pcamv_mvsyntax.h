/*
 * pcamv_mvsyntax.h -- H.264 MV-syntax extractor: host code of the product library.
 *
 * The payload sits in the motion vectors of the FINAL stream.  The reference has no extractor at all (F6); round 1 checked
 * BER = 0 on the encoder's own record (pass-1 record + flips).  This is the decode side: it parses the slice data of a
 * CABAC-coded P slice the way a decoder does -- arithmetic decoding engine (H.264 9.3.3.2), mb_skip_flag, mb_type,
 * sub_mb_type, mvd, coded_block_pattern, mb_qp_delta and the residual (which has to be decoded to keep the engine in step),
 * MV prediction (8.4.1) -- and returns every macroblock's type, partitioning and motion vectors in the record's layout, from
 * which the carrier LSBs and then the message (pcamv_gpu_stc_extract*) follow.
 *
 * Scope = what this encoder family writes in the P slices of the path: frame macroblocks, one reference picture (ref_idx is
 * not coded), 4x4 transform, cabac_init_idc 0, no intra macroblocks (the fork never picks one in a P frame):
 * anything else is reported as PCAMV_EUNSUP, a stream that does not end where it should as PCAMV_EINVAL.
 * The syntax it mirrors is the one encoder/cabac.c writes (x264_macroblock_write_cabac, x264_cabac_mb_skip, block_residual_
 * write_cabac: 403-470, 540-667, 1000-1018) and pcamv_logic.h sizes (cabac_mb_header); the tables are the standard's
 * (pcamv_entropy_tables.h: context initialisers, state transitions, rangeTabLPS in the 128-state form of common/cabac.c).
 */
#ifndef PCAMV_MVSYNTAX_H
#define PCAMV_MVSYNTAX_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace mvsyntax {

struct CabDec {
    const uint8_t *p, *end;
    uint32_t range, offset;
    int bitpos;                 /* bits of *p already consumed */
    int overrun;
    uint8_t state[464];
    int bit()
    {
        if (p >= end) { overrun++; return 0; }
        const int b = (*p >> (7 - bitpos)) & 1;
        if (++bitpos == 8) { bitpos = 0; p++; }
        return b;
    }
    void init(const uint8_t *d, size_t n, int qp)
    {
        p = d; end = d + n; bitpos = 0; overrun = 0;
        pcamv_build_cabac_init(qp, state);
        range = 510; offset = 0;
        for (int i = 0; i < 9; i++) offset = offset << 1 | (uint32_t)bit();
    }
    int decision(int ctx)
    {
        const int s = state[ctx];
        const uint32_t rlps = pcamv_cabac_range_lps[4 * s + ((range >> 6) & 3)];
        int b = s >> 6;
        range -= rlps;
        if (offset >= range) { b ^= 1; offset -= range; range = rlps; }
        state[ctx] = pcamv_cabac_transition[2 * s + b];
        while (range < 256) { range <<= 1; offset = offset << 1 | (uint32_t)bit(); }
        return b;
    }
    int bypass()
    {
        offset = offset << 1 | (uint32_t)bit();
        if (offset >= range) { offset -= range; return 1; }
        return 0;
    }
    int terminal()
    {
        range -= 2;
        if (offset >= range) return 1;
        while (range < 256) { range <<= 1; offset = offset << 1 | (uint32_t)bit(); }
        return 0;
    }
    int ue_bypass(int k)          /* Exp-Golomb suffix of UEGk (9.3.2.3) */
    {
        int v = 0;
        while (bypass()) { v += 1 << k; if (++k > 24) { overrun++; break; } }
        while (k--) v += bypass() << k;
        return v;
    }
};

enum { S8_0 = 4 + 1 * 8 };
static inline int blk_x(int idx) { return (idx & 1) | ((idx >> 1) & 2); }
static inline int blk_y(int idx) { return ((idx >> 1) & 1) | ((idx >> 2) & 2); }
static inline int s8(int idx) { return S8_0 + blk_x(idx) + 8 * blk_y(idx); }
static inline int med3(int a, int b, int c) { const int mn = a < b ? a : b, mx = a < b ? b : a; return mn > c ? mn : (mx < c ? mx : c); }

/* one macroblock's neighbourhood in x264's cache layout (8 columns; row 0 = the line above, column 3 = the column to the left) */
struct MbCache {
    int16_t mv[48][2], mvd[48][2];
    int8_t ref[48];             /* 0 = predicted from the one reference, -2 = not available (outside the picture / not decoded yet) */
    uint8_t nz[48];             /* coded_block_flag of the 4x4 blocks: luma in the motion layout; chroma see nzc_pos */
    int partition;
};
static inline void predict_from3(int ref, int refa, int refb, int refc, const int16_t *a, const int16_t *b, const int16_t *c, int mvp[2])
{
    const int cnt = (refa == ref) + (refb == ref) + (refc == ref);
    if (cnt > 1) { mvp[0] = med3(a[0], b[0], c[0]); mvp[1] = med3(a[1], b[1], c[1]); }
    else if (cnt == 1) { const int16_t *s = refa == ref ? a : refb == ref ? b : c; mvp[0] = s[0]; mvp[1] = s[1]; }
    else if (refb == -2 && refc == -2 && refa != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = med3(a[0], b[0], c[0]); mvp[1] = med3(a[1], b[1], c[1]); }
}
/* 8.4.1.3 for the partition whose first 4x4 block is idx, `width` blocks wide (the form of common/macroblock.c:165-231) */
static inline void predict_mv(const MbCache &C, int idx, int width, int mvp[2])
{
    const int i8 = s8(idx), ref = 0;
    int refa = C.ref[i8 - 1], refb = C.ref[i8 - 8], refc = C.ref[i8 - 8 + width];
    const int16_t *a = C.mv[i8 - 1], *b = C.mv[i8 - 8], *c = C.mv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || refc == -2) { refc = C.ref[i8 - 8 - 1]; c = C.mv[i8 - 8 - 1]; }
    if (C.partition == PCAMV_D_16x8) {
        if (idx == 0 && refb == ref) { mvp[0] = b[0]; mvp[1] = b[1]; return; }
        if (idx != 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
    } else if (C.partition == PCAMV_D_8x16) {
        if (idx == 0 && refa == ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
        if (idx != 0 && refc == ref) { mvp[0] = c[0]; mvp[1] = c[1]; return; }
    }
    predict_from3(ref, refa, refb, refc, a, b, c, mvp);
}
static inline void predict_pskip(const MbCache &C, int mv[2])       /* 8.4.1.1 */
{
    const int refa = C.ref[S8_0 - 1], refb = C.ref[S8_0 - 8];
    const int16_t *a = C.mv[S8_0 - 1], *b = C.mv[S8_0 - 8];
    if (refa == -2 || refb == -2 || !(refa | a[0] | a[1]) || !(refb | b[0] | b[1])) { mv[0] = mv[1] = 0; return; }
    int refc = C.ref[S8_0 - 8 + 4];
    const int16_t *c = C.mv[S8_0 - 8 + 4];
    if (refc == -2) { refc = C.ref[S8_0 - 8 - 1]; c = C.mv[S8_0 - 8 - 1]; }
    predict_from3(0, refa, refb, refc, a, b, c, mv);
}
/* position of the coded_block_flag of block idx (0..15 luma, 16..19 Cb, 20..23 Cr) in MbCache::nz: common/common.h:217-238 */
static inline int nzc_pos(int idx)
{
    if (idx < 16) return s8(idx);
    return 1 + ((idx - 16) & 1) + 8 * (1 + (((idx - 16) >> 1) & 1) + 3 * ((idx - 16) >> 2));
}

struct Parser {
    CabDec d;
    int mb_w, mb_h;
    /* what the following macroblocks read of a decoded one */
    int16_t *fmv, *fmvd;        /* [4 mb_h][4 mb_w][2] */
    uint8_t *fnz;               /* [n_mb][24] coded_block_flags, block order */
    int16_t *fcbp;              /* [n_mb] luma | chroma << 4 | chroma DC flags << 8 (Cb), << 9 (Cr) */
    int8_t *ftype;              /* [n_mb] */
    int last_dqp;

    int mvd_cpn(const MbCache &C, int idx, int l)                       /* encoder/cabac.c:403-449 */
    {
        const int i8 = s8(idx);
        const int amvd = abs(C.mvd[i8 - 1][l]) + abs(C.mvd[i8 - 8][l]);
        const int base = l ? 47 : 40;
        if (!d.decision(base + (amvd > 2) + (amvd > 32))) return 0;
        int a = 1;
        while (a < 9 && d.decision(base + (a + 2 < 6 ? a + 2 : 6))) a++;
        if (a == 9) a += d.ue_bypass(3);
        return d.bypass() ? -a : a;
    }
    void mvd(MbCache &C, int idx, int width, int height)
    {
        int mvp[2];
        predict_mv(C, idx, width, mvp);
        const int dx = mvd_cpn(C, idx, 0), dy = mvd_cpn(C, idx, 1);
        for (int j = 0; j < height; j++)
            for (int i = 0; i < width; i++) {
                const int q = s8(idx) + i + 8 * j;
                C.mv[q][0] = (int16_t)(mvp[0] + dx); C.mv[q][1] = (int16_t)(mvp[1] + dy);
                C.mvd[q][0] = (int16_t)dx; C.mvd[q][1] = (int16_t)dy; C.ref[q] = 0;
            }
    }
    /* one residual block (9.3.2.5-7 as block_residual_write_cabac writes it): returns its coded_block_flag */
    int residual(int cat, int inc)
    {
        static const int sig_off[5] = {105, 120, 134, 149, 152}, last_off[5] = {166, 181, 195, 210, 213}, lvl_off[5] = {227, 237, 247, 257, 266};
        const int cnt = cat == 3 ? 4 : cat == 4 ? 15 : 16;
        if (!d.decision(85 + 4 * cat + inc)) return 0;
        int sig[16], n = 0, i;
        for (i = 0; i < cnt - 1; i++) {
            if (d.decision(sig_off[cat] + i)) {
                sig[n++] = i;
                if (d.decision(last_off[cat] + i)) break;
            }
        }
        if (i == cnt - 1) sig[n++] = i;
        int neq1 = 0, ngt1 = 0;
        for (int k = n - 1; k >= 0; k--) {
            const int node = ngt1 ? (3 + ngt1 < 7 ? 3 + ngt1 : 7) : (neq1 < 3 ? neq1 : 3);
            const int c1 = node < 4 ? node + 1 : 0, c2 = node < 4 ? 5 : (node + 2 < 9 ? node + 2 : 9);
            if (d.decision(lvl_off[cat] + c1)) {
                int prefix = 1;
                while (prefix < 14 && d.decision(lvl_off[cat] + c2)) prefix++;
                if (prefix == 14) d.ue_bypass(0);
                ngt1++;
            } else neq1++;
            d.bypass();                                                 /* sign */
        }
        return 1;
    }

    int run(pcamv_mb_t *out)
    {
        last_dqp = 0;
        for (int my = 0; my < mb_h; my++)
            for (int mx = 0; mx < mb_w; mx++) {
                const int xy = my * mb_w + mx, s4 = 4 * mb_w;
                const bool left = mx > 0, top = my > 0, topleft = left && top, topright = top && mx < mb_w - 1;
                MbCache C;
                memset(&C, 0, sizeof(C));
                memset(C.ref, -2, sizeof(C.ref));
                C.partition = PCAMV_D_16x16;
                auto take = [&](int q, int bx, int by) {                /* neighbour 4x4 block (bx, by) of the picture into cache position q */
                    const int16_t *m = fmv + 2 * (by * s4 + bx), *dd = fmvd + 2 * (by * s4 + bx);
                    C.mv[q][0] = m[0]; C.mv[q][1] = m[1]; C.mvd[q][0] = dd[0]; C.mvd[q][1] = dd[1]; C.ref[q] = 0;
                };
                if (left) for (int j = 0; j < 4; j++) take(S8_0 - 1 + 8 * j, 4 * mx - 1, 4 * my + j);
                if (top) for (int i = 0; i < 4; i++) take(S8_0 - 8 + i, 4 * mx + i, 4 * my - 1);
                if (topleft) take(S8_0 - 8 - 1, 4 * mx - 1, 4 * my - 1);
                if (topright) take(S8_0 - 8 + 4, 4 * mx + 4, 4 * my - 1);
                /* coded_block_flags of the neighbours: not available counts as 0 for an inter macroblock (9.3.3.1.1.9) */
                if (left) { const uint8_t *z = fnz + 24 * (xy - 1);
                    C.nz[nzc_pos(0) - 1] = z[5]; C.nz[nzc_pos(2) - 1] = z[7]; C.nz[nzc_pos(8) - 1] = z[13]; C.nz[nzc_pos(10) - 1] = z[15];
                    C.nz[nzc_pos(16) - 1] = z[17]; C.nz[nzc_pos(18) - 1] = z[19]; C.nz[nzc_pos(20) - 1] = z[21]; C.nz[nzc_pos(22) - 1] = z[23]; }
                if (top) { const uint8_t *z = fnz + 24 * (xy - mb_w);
                    C.nz[nzc_pos(0) - 8] = z[10]; C.nz[nzc_pos(1) - 8] = z[11]; C.nz[nzc_pos(4) - 8] = z[14]; C.nz[nzc_pos(5) - 8] = z[15];
                    C.nz[nzc_pos(16) - 8] = z[18]; C.nz[nzc_pos(17) - 8] = z[19]; C.nz[nzc_pos(20) - 8] = z[22]; C.nz[nzc_pos(21) - 8] = z[23]; }
                const int cl = left ? fcbp[xy - 1] : -1, ct = top ? fcbp[xy - mb_w] : -1;
                const int tl = left ? ftype[xy - 1] : -1, tt = top ? ftype[xy - mb_w] : -1;

                pcamv_mb_t *o = &out[xy];
                memset(o, 0, sizeof(*o));
                for (int i = 0; i < 4; i++) o->i_sub_partition[i] = PCAMV_D_L0_8x8;
                o->i_partition = PCAMV_D_16x16;
                uint8_t sub[4] = {PCAMV_D_L0_8x8, PCAMV_D_L0_8x8, PCAMV_D_L0_8x8, PCAMV_D_L0_8x8};
                int cbp_luma = 0, cbp_chroma = 0, dcf = 0;
                uint8_t nzb[24];
                memset(nzb, 0, sizeof(nzb));
                const int skip = d.decision(11 + (tl >= 0 && tl != PCAMV_P_SKIP) + (tt >= 0 && tt != PCAMV_P_SKIP));
                if (skip) {
                    int mv[2];
                    predict_pskip(C, mv);
                    for (int i = 0; i < 16; i++) { C.mv[s8(i)][0] = (int16_t)mv[0]; C.mv[s8(i)][1] = (int16_t)mv[1]; C.ref[s8(i)] = 0; }
                    o->i_type = PCAMV_P_SKIP;
                    o->pskip_mv[0] = (int16_t)mv[0]; o->pskip_mv[1] = (int16_t)mv[1];
                    last_dqp = 0;
                } else {
                    if (d.decision(14)) return PCAMV_EUNSUP;            /* an intra macroblock in a P slice */
                    if (!d.decision(15)) { if (d.decision(16)) { o->i_type = PCAMV_P_8x8; o->i_partition = PCAMV_D_8x8; } else o->i_type = PCAMV_P_L0; }
                    else { o->i_type = PCAMV_P_L0; o->i_partition = d.decision(17) ? PCAMV_D_16x8 : PCAMV_D_8x16; }
                    C.partition = o->i_partition;
                    if (o->i_type == PCAMV_P_8x8) {
                        for (int i = 0; i < 4; i++) {
                            if (d.decision(21)) sub[i] = PCAMV_D_L0_8x8;
                            else if (!d.decision(22)) sub[i] = PCAMV_D_L0_8x4;
                            else sub[i] = d.decision(23) ? PCAMV_D_L0_4x8 : PCAMV_D_L0_4x4;
                        }
                        for (int i = 0; i < 4; i++)
                            switch (sub[i]) {
                            case PCAMV_D_L0_8x8: mvd(C, 4 * i, 2, 2); break;
                            case PCAMV_D_L0_8x4: mvd(C, 4 * i, 2, 1); mvd(C, 4 * i + 2, 2, 1); break;
                            case PCAMV_D_L0_4x8: mvd(C, 4 * i, 1, 2); mvd(C, 4 * i + 1, 1, 2); break;
                            default: for (int k = 0; k < 4; k++) mvd(C, 4 * i + k, 1, 1); break;
                            }
                        memcpy(o->i_sub_partition, sub, 4);
                    } else if (o->i_partition == PCAMV_D_16x16) mvd(C, 0, 4, 4);
                    else if (o->i_partition == PCAMV_D_16x8) { mvd(C, 0, 4, 2); mvd(C, 8, 4, 2); }
                    else { mvd(C, 0, 2, 4); mvd(C, 4, 2, 4); }
                    /* coded_block_pattern (encoder/cabac.c:300-356) */
                    int b;
                    b = d.decision(76 - ((cl >> 1) & 1) - ((ct >> 1) & 2)); cbp_luma |= b;
                    b = d.decision(76 - (cbp_luma & 1) - ((ct >> 2) & 2)); cbp_luma |= b << 1;
                    b = d.decision(76 - ((cl >> 3) & 1) - ((cbp_luma << 1) & 2)); cbp_luma |= b << 2;
                    b = d.decision(76 - ((cbp_luma >> 2) & 1) - (cbp_luma & 2)); cbp_luma |= b << 3;
                    const int ca = cl & 0x30, cb = ct & 0x30;
                    if (d.decision(77 + ((ca && cl != -1) ? 1 : 0) + ((cb && ct != -1) ? 2 : 0)))
                        cbp_chroma = 1 + d.decision(77 + 4 + (ca == 0x20) + 2 * (cb == 0x20));
                    if (cbp_luma | cbp_chroma) {                        /* mb_qp_delta */
                        int n = 0;
                        if (d.decision(60 + (last_dqp != 0))) { n = 1; while (d.decision(n == 1 ? 62 : 63)) if (++n > 104) return PCAMV_EINVAL; }
                        last_dqp = n ? ((n + 1) >> 1) * ((n & 1) ? 1 : -1) : 0;
                        for (int i = 0; i < 16; i++)
                            if ((cbp_luma >> (i >> 2)) & 1) {
                                const int q = nzc_pos(i);
                                nzb[i] = (uint8_t)residual(2, (C.nz[q - 1] != 0) + 2 * (C.nz[q - 8] != 0));
                                C.nz[q] = nzb[i];
                            }
                        if (cbp_chroma) {
                            for (int k = 0; k < 2; k++) {
                                const int inc = (cl != -1 ? (cl >> (8 + k)) & 1 : 0) + 2 * (ct != -1 ? (ct >> (8 + k)) & 1 : 0);
                                dcf |= residual(3, inc) << k;
                            }
                            if (cbp_chroma == 2)
                                for (int i = 16; i < 24; i++) {
                                    const int q = nzc_pos(i);
                                    nzb[i] = (uint8_t)residual(4, (C.nz[q - 1] != 0) + 2 * (C.nz[q - 8] != 0));
                                    C.nz[q] = nzb[i];
                                }
                        }
                    } else last_dqp = 0;
                }
                /* what the next macroblocks read */
                for (int i = 0; i < 16; i++) {
                    const int bx = 4 * mx + blk_x(i), by = 4 * my + blk_y(i), q = s8(i);
                    fmv[2 * (by * s4 + bx)] = C.mv[q][0]; fmv[2 * (by * s4 + bx) + 1] = C.mv[q][1];
                    fmvd[2 * (by * s4 + bx)] = skip ? 0 : C.mvd[q][0]; fmvd[2 * (by * s4 + bx) + 1] = skip ? 0 : C.mvd[q][1];
                    o->mv[i][0] = C.mv[q][0]; o->mv[i][1] = C.mv[q][1]; o->ref[i] = 0;
                }
                memcpy(fnz + 24 * xy, nzb, 24);
                fcbp[xy] = (int16_t)(cbp_luma | cbp_chroma << 4 | dcf << 8);
                ftype[xy] = (int8_t)o->i_type;
                const int end = d.terminal();
                if (end != (xy == mb_w * mb_h - 1)) return PCAMV_EINVAL;       /* end_of_slice_flag in the wrong place */
                if (d.overrun) return PCAMV_EINVAL;
            }
        return 0;
    }
};

/* ---------------------------------------------------------------- CAVLC form (encoder/cavlc.c) */
struct BitRd {
    const uint8_t *d; size_t nbits, pos; int overrun;
    unsigned peek(int n) const        /* the next n <= 24 bits, zero-filled past the end */
    {
        unsigned v = 0;
        for (int i = 0; i < n; i++) { const size_t q = pos + (size_t)i; v = v << 1 | (q < nbits ? (unsigned)((d[q >> 3] >> (7 - (q & 7))) & 1) : 0u); }
        return v;
    }
    unsigned get(int n) { const unsigned v = peek(n); pos += (size_t)n; if (pos > nbits) overrun++; return v; }
    unsigned ue()
    {
        int z = 0;
        while (!get(1)) if (++z > 24 || overrun) { overrun++; return 0; }
        return ((1u << z) - 1u) + (z ? get(z) : 0u);
    }
    int se() { const unsigned k = ue(); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
};

struct ParserV {
    BitRd b;
    int mb_w, mb_h;
    int16_t *fmv; uint8_t *fnz; int8_t *ftype;      /* motion, total_coeff per 4x4 block [n_mb][24], types */
    uint8_t cbp_of[48];                             /* codeNum -> coded_block_pattern (inverse of Table 9-4's inter column) */

    /* one residual block (9.2): returns TotalCoeff.  tab: 0..3 by nC, 4 = chroma DC; maxc: 16, 15 or 4 */
    int residual(int tab, int maxc)
    {
        int total = -1, t1 = 0;
        if (b.peek(pcamv_vlc_coeff0_len[tab]) == pcamv_vlc_coeff0_code[tab]) { b.get(pcamv_vlc_coeff0_len[tab]); return 0; }
        for (int tc = 1; tc <= (tab == 4 ? 4 : 16) && total < 0; tc++)
            for (int tr = 0; tr <= (tc < 3 ? tc : 3); tr++) {
                const int k = tab * 64 + (tc - 1) * 4 + tr, len = pcamv_vlc_coeff_len[k];
                if (len && b.peek(len) == pcamv_vlc_coeff_code[k]) { b.get(len); total = tc; t1 = tr; break; }
            }
        if (total < 0 || total > maxc) { b.overrun++; return 0; }
        int suffix_len = total > 10 && t1 < 3;
        b.get(t1);                                                       /* signs of the trailing ones */
        for (int i = t1; i < total; i++) {
            int prefix = 0;
            while (!b.get(1)) if (++prefix > 31 || b.overrun) { b.overrun++; return 0; }
            const int ssize = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
            int code = ((prefix < 15 ? prefix : 15) << suffix_len) + (ssize ? (int)b.get(ssize) : 0);
            if (prefix >= 15 && suffix_len == 0) code += 15;
            if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
            if (i == t1 && t1 < 3) code += 2;
            const int a = (code + 2) >> 1;                              /* |level| */
            if (suffix_len == 0) suffix_len = 1;
            if (a > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
        int zeros = 0;
        if (total < maxc) {
            const unsigned char *len = tab == 4 ? &pcamv_vlc_total_zeros_dc_len[(total - 1) * 4] : &pcamv_vlc_total_zeros_len[(total - 1) * 16];
            const unsigned short *code = tab == 4 ? &pcamv_vlc_total_zeros_dc_code[(total - 1) * 4] : &pcamv_vlc_total_zeros_code[(total - 1) * 16];
            int z, nz = tab == 4 ? 4 : 16;
            for (z = 0; z < nz; z++) if (len[z] && b.peek(len[z]) == code[z]) { b.get(len[z]); break; }
            if (z == nz) { b.overrun++; return 0; }
            zeros = z;
        }
        for (int i = 0; i < total - 1 && zeros > 0; i++) {
            const int zl = zeros - 1 < 6 ? zeros - 1 : 6;
            int r;
            for (r = 0; r < 16; r++) { const int len = pcamv_vlc_run_before_len[zl * 16 + r]; if (len && b.peek(len) == pcamv_vlc_run_before_code[zl * 16 + r]) { b.get(len); break; } }
            if (r == 16 || r > zeros) { b.overrun++; return 0; }
            zeros -= r;
        }
        return total;
    }
    void mvd(MbCache &C, int idx, int width, int height)
    {
        int mvp[2];
        predict_mv(C, idx, width, mvp);
        const int dx = b.se(), dy = b.se();
        for (int j = 0; j < height; j++)
            for (int i = 0; i < width; i++) {
                const int q = s8(idx) + i + 8 * j;
                C.mv[q][0] = (int16_t)(mvp[0] + dx); C.mv[q][1] = (int16_t)(mvp[1] + dy); C.ref[q] = 0;
            }
    }
    int run(pcamv_mb_t *out)
    {
        static const uint8_t ct_index[17] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3};
        for (int i = 0; i < 48; i++) cbp_of[pcamv_inter_cbp_to_golomb[i]] = (uint8_t)i;
        int skip_run = -1;              /* -1: the next thing in the stream is an mb_skip_run */
        for (int my = 0; my < mb_h; my++)
            for (int mx = 0; mx < mb_w; mx++) {
                const int xy = my * mb_w + mx, s4 = 4 * mb_w;
                const bool left = mx > 0, top = my > 0, topleft = left && top, topright = top && mx < mb_w - 1;
                MbCache C;
                memset(&C, 0, sizeof(C));
                memset(C.ref, -2, sizeof(C.ref));
                memset(C.nz, 0x80, sizeof(C.nz));                         /* 0x80: not available (common/macroblock.c:1100-1160) */
                C.partition = PCAMV_D_16x16;
                auto take = [&](int q, int bx, int by) { const int16_t *m = fmv + 2 * (by * s4 + bx); C.mv[q][0] = m[0]; C.mv[q][1] = m[1]; C.ref[q] = 0; };
                if (left) for (int j = 0; j < 4; j++) take(S8_0 - 1 + 8 * j, 4 * mx - 1, 4 * my + j);
                if (top) for (int i = 0; i < 4; i++) take(S8_0 - 8 + i, 4 * mx + i, 4 * my - 1);
                if (topleft) take(S8_0 - 8 - 1, 4 * mx - 1, 4 * my - 1);
                if (topright) take(S8_0 - 8 + 4, 4 * mx + 4, 4 * my - 1);
                if (left) { const uint8_t *z = fnz + 24 * (xy - 1);
                    C.nz[nzc_pos(0) - 1] = z[5]; C.nz[nzc_pos(2) - 1] = z[7]; C.nz[nzc_pos(8) - 1] = z[13]; C.nz[nzc_pos(10) - 1] = z[15];
                    C.nz[nzc_pos(16) - 1] = z[17]; C.nz[nzc_pos(18) - 1] = z[19]; C.nz[nzc_pos(20) - 1] = z[21]; C.nz[nzc_pos(22) - 1] = z[23]; }
                if (top) { const uint8_t *z = fnz + 24 * (xy - mb_w);
                    C.nz[nzc_pos(0) - 8] = z[10]; C.nz[nzc_pos(1) - 8] = z[11]; C.nz[nzc_pos(4) - 8] = z[14]; C.nz[nzc_pos(5) - 8] = z[15];
                    C.nz[nzc_pos(16) - 8] = z[18]; C.nz[nzc_pos(17) - 8] = z[19]; C.nz[nzc_pos(20) - 8] = z[22]; C.nz[nzc_pos(21) - 8] = z[23]; }
                for (int i = 0; i < 24; i++) C.nz[nzc_pos(i)] = 0;
                pcamv_mb_t *o = &out[xy];
                memset(o, 0, sizeof(*o));
                for (int i = 0; i < 4; i++) o->i_sub_partition[i] = PCAMV_D_L0_8x8;
                o->i_partition = PCAMV_D_16x16;
                uint8_t nzb[24];
                memset(nzb, 0, sizeof(nzb));
                if (skip_run < 0) skip_run = (int)b.ue();
                if (skip_run > 0) {
                    skip_run--;
                    int mv[2];
                    predict_pskip(C, mv);
                    for (int i = 0; i < 16; i++) { C.mv[s8(i)][0] = (int16_t)mv[0]; C.mv[s8(i)][1] = (int16_t)mv[1]; }
                    o->i_type = PCAMV_P_SKIP;
                    o->pskip_mv[0] = (int16_t)mv[0]; o->pskip_mv[1] = (int16_t)mv[1];
                    /* (when the run is used up the next macroblock is a coded one: no new run is read before it) */
                } else {
                    skip_run = -1;
                    const unsigned mt = b.ue();
                    if (mt > 4) return PCAMV_EUNSUP;                    /* an intra macroblock in a P slice */
                    uint8_t sub[4] = {PCAMV_D_L0_8x8, PCAMV_D_L0_8x8, PCAMV_D_L0_8x8, PCAMV_D_L0_8x8};
                    if (mt >= 3) {
                        static const uint8_t sub_of[4] = {PCAMV_D_L0_8x8, PCAMV_D_L0_8x4, PCAMV_D_L0_4x8, PCAMV_D_L0_4x4};
                        o->i_type = PCAMV_P_8x8; o->i_partition = PCAMV_D_8x8; C.partition = PCAMV_D_8x8;
                        for (int i = 0; i < 4; i++) { const unsigned st = b.ue(); if (st > 3) return PCAMV_EINVAL; sub[i] = sub_of[st]; }
                        for (int i = 0; i < 4; i++)
                            switch (sub[i]) {
                            case PCAMV_D_L0_8x8: mvd(C, 4 * i, 2, 2); break;
                            case PCAMV_D_L0_8x4: mvd(C, 4 * i, 2, 1); mvd(C, 4 * i + 2, 2, 1); break;
                            case PCAMV_D_L0_4x8: mvd(C, 4 * i, 1, 2); mvd(C, 4 * i + 1, 1, 2); break;
                            default: for (int k = 0; k < 4; k++) mvd(C, 4 * i + k, 1, 1); break;
                            }
                        memcpy(o->i_sub_partition, sub, 4);
                    } else {
                        o->i_type = PCAMV_P_L0;
                        o->i_partition = mt == 0 ? PCAMV_D_16x16 : mt == 1 ? PCAMV_D_16x8 : PCAMV_D_8x16;
                        C.partition = o->i_partition;
                        if (mt == 0) mvd(C, 0, 4, 4);
                        else if (mt == 1) { mvd(C, 0, 4, 2); mvd(C, 8, 4, 2); }
                        else { mvd(C, 0, 2, 4); mvd(C, 4, 2, 4); }
                    }
                    const unsigned cn = b.ue();
                    if (cn > 47) return PCAMV_EINVAL;
                    const int cbp = cbp_of[cn], cbp_luma = cbp & 15, cbp_chroma = cbp >> 4;
                    if (cbp) {
                        b.se();                                         /* mb_qp_delta */
                        for (int i = 0; i < 16; i++)
                            if ((cbp_luma >> (i >> 2)) & 1) {
                                const int q = nzc_pos(i);
                                int nc = C.nz[q - 1] + C.nz[q - 8];
                                if (nc < 0x80) nc = (nc + 1) >> 1;
                                nzb[i] = (uint8_t)residual(ct_index[nc & 0x7f], 16);
                                C.nz[q] = nzb[i];
                            }
                        if (cbp_chroma) {
                            residual(4, 4); residual(4, 4);
                            if (cbp_chroma & 2)
                                for (int i = 16; i < 24; i++) {
                                    const int q = nzc_pos(i);
                                    int nc = C.nz[q - 1] + C.nz[q - 8];
                                    if (nc < 0x80) nc = (nc + 1) >> 1;
                                    nzb[i] = (uint8_t)residual(ct_index[nc & 0x7f], 15);
                                    C.nz[q] = nzb[i];
                                }
                        }
                    }
                }
                for (int i = 0; i < 16; i++) {
                    const int bx = 4 * mx + blk_x(i), by = 4 * my + blk_y(i), q = s8(i);
                    fmv[2 * (by * s4 + bx)] = C.mv[q][0]; fmv[2 * (by * s4 + bx) + 1] = C.mv[q][1];
                    o->mv[i][0] = C.mv[q][0]; o->mv[i][1] = C.mv[q][1]; o->ref[i] = 0;
                }
                memcpy(fnz + 24 * xy, nzb, 24);
                ftype[xy] = (int8_t)o->i_type;
                if (b.overrun) return PCAMV_EINVAL;
            }
        if (skip_run > 0) return PCAMV_EINVAL;       /* the last mb_skip_run claims more macroblocks than the picture has left */
        /* rbsp_slice_trailing_bits: a 1 and zeros up to the byte boundary, within the last byte(s) handed over */
        if (b.pos >= b.nbits || !b.get(1)) return PCAMV_EINVAL;
        while (b.pos < b.nbits) if (b.get(1)) return PCAMV_EINVAL;
        return 0;
    }
};
}   /* namespace mvsyntax */

extern "C" int pcamv_gpu_parse_pslice_cabac(const uint8_t *data, size_t len, int mb_w, int mb_h, int slice_qp, pcamv_mb_t *out_mb)
{
    if (!data || !out_mb || len < 2 || mb_w < 1 || mb_h < 1 || slice_qp < 0 || slice_qp > 51) return PCAMV_EINVAL;
    const size_t n = (size_t)mb_w * mb_h;
    mvsyntax::Parser P;
    P.mb_w = mb_w; P.mb_h = mb_h;
    P.fmv = (int16_t *)calloc(n * 32, sizeof(int16_t)); P.fmvd = (int16_t *)calloc(n * 32, sizeof(int16_t));
    P.fnz = (uint8_t *)calloc(n * 24, 1); P.fcbp = (int16_t *)calloc(n, sizeof(int16_t)); P.ftype = (int8_t *)calloc(n, 1);
    int rc = PCAMV_ENOMEM;
    if (P.fmv && P.fmvd && P.fnz && P.fcbp && P.ftype) {
        P.d.init(data, len, slice_qp);
        rc = P.run(out_mb);
    }
    free(P.fmv); free(P.fmvd); free(P.fnz); free(P.fcbp); free(P.ftype);
    return rc;
}
/* The slice data where a real stream has it: inside a NAL unit (emulation prevention bytes, x264_nal_encode common/common.c:658-690)
 * and behind a slice header of any bit length.  pcamv_gpu_nal_to_rbsp undoes the NAL layer; the _at forms start at a bit of the
 * RBSP: CAVLC slice data follows the header directly, CABAC slice data after cabac_alignment_one_bits up to the byte boundary. */
extern "C" int pcamv_gpu_nal_to_rbsp(const uint8_t *nal, size_t len, uint8_t *rbsp, size_t *rbsp_len, int *nal_ref_idc, int *nal_unit_type)
{
    if (!nal || !rbsp || !rbsp_len) return PCAMV_EINVAL;
    size_t i = 0;
    if (len >= 4 && nal[0] == 0 && nal[1] == 0 && nal[2] == 0 && nal[3] == 1) i = 4;           /* Annex B start code, long or short */
    else if (len >= 3 && nal[0] == 0 && nal[1] == 0 && nal[2] == 1) i = 3;
    if (i >= len || (nal[i] & 0x80)) return PCAMV_EINVAL;                                       /* forbidden_zero_bit */
    if (nal_ref_idc) *nal_ref_idc = (nal[i] >> 5) & 3;
    if (nal_unit_type) *nal_unit_type = nal[i] & 31;
    i++;
    size_t n = 0; int zeros = 0;
    for (; i < len; i++) {
        if (zeros >= 2 && nal[i] == 3) {            /* emulation_prevention_three_byte: dropped */
            if (i + 1 < len && nal[i + 1] > 3) return PCAMV_EINVAL;     /* 00 00 03 xx with xx > 03 does not occur in a NAL unit */
            zeros = 0;
            continue;
        }
        if (zeros >= 2 && nal[i] < 3) return PCAMV_EINVAL;              /* 00 00 00 / 01 / 02 inside a NAL unit: the next start code, not payload */
        zeros = nal[i] == 0 ? zeros + 1 : 0;
        rbsp[n++] = nal[i];
    }
    *rbsp_len = n;
    return 0;
}
extern "C" int pcamv_gpu_parse_pslice_cabac(const uint8_t *data, size_t len, int mb_w, int mb_h, int slice_qp, pcamv_mb_t *out_mb);
extern "C" int pcamv_gpu_parse_pslice_cabac_at(const uint8_t *rbsp, size_t len, size_t start_bit, int mb_w, int mb_h, int slice_qp, pcamv_mb_t *out_mb)
{
    if (!rbsp || start_bit > len * 8) return PCAMV_EINVAL;
    while (start_bit & 7) {                         /* cabac_alignment_one_bit */
        if (!((rbsp[start_bit >> 3] >> (7 - (start_bit & 7))) & 1)) return PCAMV_EINVAL;
        start_bit++;
    }
    return pcamv_gpu_parse_pslice_cabac(rbsp + (start_bit >> 3), len - (start_bit >> 3), mb_w, mb_h, slice_qp, out_mb);
}
static int parse_pslice_cavlc_bits(const uint8_t *data, size_t len, size_t start_bit, int mb_w, int mb_h, pcamv_mb_t *out_mb);
extern "C" int pcamv_gpu_parse_pslice_cavlc_at(const uint8_t *rbsp, size_t len, size_t start_bit, int mb_w, int mb_h, pcamv_mb_t *out_mb)
{
    if (!rbsp || start_bit >= len * 8) return PCAMV_EINVAL;
    return parse_pslice_cavlc_bits(rbsp, len, start_bit, mb_w, mb_h, out_mb);
}
extern "C" int pcamv_gpu_parse_pslice_cavlc(const uint8_t *data, size_t len, int mb_w, int mb_h, pcamv_mb_t *out_mb)
{
    return parse_pslice_cavlc_bits(data, len, 0, mb_w, mb_h, out_mb);
}
static int parse_pslice_cavlc_bits(const uint8_t *data, size_t len, size_t start_bit, int mb_w, int mb_h, pcamv_mb_t *out_mb)
{
    if (!data || !out_mb || len < 1 || mb_w < 1 || mb_h < 1) return PCAMV_EINVAL;
    const size_t n = (size_t)mb_w * mb_h;
    mvsyntax::ParserV P;
    P.mb_w = mb_w; P.mb_h = mb_h;
    P.fmv = (int16_t *)calloc(n * 32, sizeof(int16_t)); P.fnz = (uint8_t *)calloc(n * 24, 1); P.ftype = (int8_t *)calloc(n, 1);
    int rc = PCAMV_ENOMEM;
    if (P.fmv && P.fnz && P.ftype) {
        P.b.d = data; P.b.nbits = len * 8; P.b.pos = start_bit; P.b.overrun = 0;
        rc = P.run(out_mb);
    }
    free(P.fmv); free(P.fnz); free(P.ftype);
    return rc;
}
#endif
