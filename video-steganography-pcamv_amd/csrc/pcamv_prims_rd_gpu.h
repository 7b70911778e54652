/*
 * pcamv_prims_rd_gpu.h -- lane-parallel primitives of the RD mode decision (--subme 6 / 7), one wavefront = one macroblock.
 *
 *   intra SATD analysis     common/predict.c (16x16: 4 modes x 16 blocks = 64 lanes; chroma: 4 modes x 2 planes x 4 blocks;
 *                           4x4: one mode per lane), encoder/macroblock.c:116-148 (the 4x4 blocks' own reconstruction)
 *   psy-RD / SSD            encoder/rdo.c:65-137, encoder/analyse.c:522-549, common/pixel.c:256-358: 4x4 Hadamard per lane,
 *                           the 8x8 transform as a 2x2 Hadamard across the four lanes of a quad
 *   CABAC size walk         encoder/cabac.c:540-667: per block the significance map is one decision per lane (the lane that
 *                           owns scan position i keeps the states of "significant[i]" / "last[i]" in registers across the
 *                           blocks of a category), the level chain runs over the non-zero levels only, each decision on the
 *                           lane that owns that level context
 *   CAVLC size              encoder/cavlc.c:109-199: one block per lane
 */
#ifndef PCAMV_PRIMS_RD_GPU_H
#define PCAMV_PRIMS_RD_GPU_H

__device__ __forceinline__ uint32_t rep4(int v) { return (uint32_t)(v & 255) * 0x01010101u; }
__device__ __forceinline__ int bsum4(uint32_t v) { return (int)__builtin_amdgcn_sad_u8(v, 0u, 0u); }

/* Everything a macroblock needs from memory before its analysis can start, apart from the neighbours' motion (pcamv_logic.h
 * mb_load): the source pixels and, for the RD decision, the intra prediction borders (unfiltered pass-1 reconstruction of the
 * neighbours), the entropy coder's context inputs (non-zero flags / counts, coded block patterns, MV differences), the slice's
 * CABAC states and the (bits, next state) table.  Two halves: prim_mb_fetch ISSUES every load into registers and uses none of
 * them, prim_mb_fetch_store puts them into LDS -- so that all of it, together with the caller's own neighbour loads, is ONE
 * memory round trip on the macroblock chain.  (One load + LDS store per role and branch used to be a dozen round trips in a row:
 * a store needs its value, so every branch waited for its own load.) */
struct MbFetch { uint32_t fy, fc, role, s0, s1, s2, t0, t1, t2, t3; int b0, b1; };
__device__ __forceinline__ void prim_mb_fetch(const FrameDev &F, int mb_x_, int mb_y_, int nb_, int rd_, MbFetch &P)
{
    const int mb_x = rfl(mb_x_), mb_y = rfl(mb_y_), nb = rfl(nb_), rd = rfl(rd_);
    const int lane = LANE();
    const int xy = mb_y * F.mb_w + mb_x, top = xy - F.mb_w;
    { const int row = lane >> 2, c4 = lane & 3;
      P.fy = *(const uint32_t *)(F.fenc[0] + (size_t)(mb_y * 16 + row) * F.w + mb_x * 16 + c4 * 4); }
    P.fc = 0;
    if (lane < 32) {
        const int plane = lane >> 4, row = (lane & 15) >> 1, c4 = lane & 1;
        P.fc = *(const uint32_t *)((plane ? F.fenc[2] : F.fenc[1]) + (size_t)(mb_y * 8 + row) * (F.w >> 1) + mb_x * 8 + c4 * 4);
    }
    P.role = 0; P.b0 = P.b1 = 0; P.s0 = P.s1 = P.s2 = P.t0 = P.t1 = P.t2 = P.t3 = 0;
    if (!rd) return;
    /* every lane makes all three loads (its role's from the neighbour, the others' from this macroblock's own slots, ignored) and keeps
     * its role's result: loads inside the role branches each waited for their own data -- a round trip per role */
    {
        const int k8 = lane & 7, l8 = (lane >> 3) & 1, ok8 = lane < 16 && (l8 ? (nb & NB_LEFT) : (nb & NB_TOP));
        const int k32 = (lane - 16) & 7, l32 = k32 >> 2, ok32 = lane >= 16 && lane < 24 && (l32 ? (nb & NB_LEFT) : (nb & NB_TOP));
        const int ok16 = (lane == 24 && (nb & NB_TOP)) || (lane == 25 && (nb & NB_LEFT));
        const uint32_t v8 = (uint8_t)NB_LD8(&F.nb_nz[(ok8 ? (l8 ? xy - 1 : top) : xy) * 16 + 8 * l8 + k8]);
        const uint32_t v32 = NB_LD32(&F.nb_mvd[((ok32 ? (l32 ? xy - 1 : top) : xy) * 8 + k32) * 2]);
        const uint32_t v16 = (uint32_t)(int)(int16_t)NB_LD16(&F.nb_cbp[ok16 ? (lane == 24 ? top : xy - 1) : xy]);
        P.role = lane < 16 ? (ok8 ? v8 : 0x80u) : ok32 ? v32 : ok16 ? v16 : 0u;
    }
    /* borders: luma 25 + 16, chroma 2 x (9 + 8) = 75 bytes, agent-scope loads (the neighbours stored them write-through) */
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int i = imin(lane + 64 * r, 74);
        int c, is_left, k;
        if (i < 41) { c = 0; is_left = i >= 25; k = is_left ? i - 25 : i - 1; }
        else { const int j = (i - 41) % 17; c = 1 + (i - 41) / 17; is_left = j >= 9; k = is_left ? j - 9 : j - 1; }
        const int w = c ? 8 : 16, pw = c ? F.w >> 1 : F.w, x0 = mb_x * w, y0 = mb_y * w;
        const uint8_t *pl = c == 0 ? F.rec[0] : c == 1 ? F.rec[1] : F.rec[2];
        const bool ok = lane + 64 * r < 75 && (is_left ? mb_x > 0 : mb_y > 0);
        const size_t o = !ok ? (size_t)y0 * pw + x0 : is_left ? (size_t)(y0 + k) * pw + x0 - 1 : (size_t)(y0 - 1) * pw + clip3i(x0 + k, 0, pw - 1);
        const int v = (uint8_t)NB_LD8((const int8_t *)pl + o);
        if (r == 0) P.b0 = ok ? v : 0; else P.b1 = ok ? v : 0;
    }
    if (F.b_cabac) {
        const uint32_t *src = (const uint32_t *)(xy == 0 ? F.cabac_init : F.cabac);
        if (xy == 0) { P.s0 = src[lane]; if (lane < 52) P.s1 = src[64 + lane]; }
        else { P.s0 = NB_LD32(src + lane); if (lane < 52) P.s1 = NB_LD32(src + 64 + lane); }
        P.t0 = F.cabac_tab[lane]; P.t1 = F.cabac_tab[64 + lane]; P.t2 = F.cabac_tab[128 + lane]; P.t3 = F.cabac_tab[192 + lane];
    }
    if (F.inter & PCAMV_ANALYSE_PSUB8x8) {       /* the macroblock's own non-zero / mvd entries as the macroblock coded before it left them (PCAMV_CHAIN_NZ / _MVD) */
        const uint32_t *src = (const uint32_t *)(xy == 0 ? F.cabac_init : F.cabac) + PCAMV_CHAIN_NZ / 4;
        if (lane < 22) P.s2 = xy == 0 ? src[lane] : NB_LD32(src + lane);
    }
}
__device__ __forceinline__ void prim_mb_fetch_store(const FrameDev &F, MBLocal *L, int rd_, const MbFetch &P)
{
    const int rd = rfl(rd_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    { const int row = lane >> 2, c4 = lane & 3; sts4(L->fenc + row * 16 + c4 * 4, P.fy); }
    if (lane < 32) { const int plane = lane >> 4, row = (lane & 15) >> 1, c4 = lane & 1; sts4(L->fenc + 256 + row * 16 + plane * 8 + c4 * 4, P.fc); }
    if (rd) {
        const int xy = L->mb_xy, nb = L->neighbour;
        if (lane < 48) { L->nzc[lane] = 0; L->i4mode[lane] = -1; ((uint32_t *)L->cmvd)[lane] = 0; }
        PCAMV_WAVE_SYNC();
        if (F.inter & PCAMV_ANALYSE_PSUB8x8) {
            if (lane < 6) {
#pragma unroll
                for (int k = 0; k < 4; k++) L->nzc[scan8_all_of(4 * lane + k)] = (uint8_t)(P.s2 >> (8 * k));
            } else if (lane < 22) ((uint32_t *)L->cmvd)[scan8_of(lane - 6)] = P.s2;
        }
        if (lane < 16) {
            const int k = lane & 7, is_left = lane >> 3;
            /* cache positions of the bottom row / right column entries: 4 luma, 2 Cb, 2 Cr */
            const int pos = is_left ? (k < 4 ? 3 + 8 * (1 + k) : k < 6 ? 0 + 8 * (1 + (k - 4)) : 0 + 8 * (4 + (k - 6)))
                                    : (k < 4 ? 4 + k : k < 6 ? 1 + (k - 4) : 1 + (k - 6) + 3 * 8);
            L->nzc[pos] = (uint8_t)P.role;
        } else if (lane < 24) {
            const int k = lane - 16, is_left = k >> 2, j = k & 3;
            if (is_left ? (nb & NB_LEFT) : (nb & NB_TOP)) {
                const int pos = is_left ? SCAN8_0 - 1 + 8 * j : SCAN8_0 - 8 + j;
                ((uint32_t *)L->cmvd)[pos] = P.role;
                L->i4mode[pos] = 2;                                  /* inter neighbours count as DC (common/macroblock.c:1282) */
            }
        } else if (lane == 24) L->cbp_top = (nb & NB_TOP) ? (int)P.role : -1;
        else if (lane == 25) L->cbp_left = (nb & NB_LEFT) ? (int)P.role : -1;
        else if (lane == 26) L->b_fast_intra = xy > 4 && F.ref_is_inter;          /* analyse.c:363-378 */
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = lane + 64 * r;
            if (i < 75) {
                int c, is_left, k;
                if (i < 41) { c = 0; is_left = i >= 25; k = is_left ? i - 25 : i - 1; }
                else { const int j = (i - 41) % 17; c = 1 + (i - 41) / 17; is_left = j >= 9; k = is_left ? j - 9 : j - 1; }
                const uint8_t v = (uint8_t)(r == 0 ? P.b0 : P.b1);
                if (is_left) L->ib_left[c][k] = v; else L->ib_top[c][4 + k] = v;
            }
        }
        if (F.b_cabac) {
            uint32_t *dst = (uint32_t *)L_CAB(L, 0);
            dst[lane] = P.s0; if (lane < 52) dst[64 + lane] = P.s1;
            L_CTAB(L)[lane] = P.t0; L_CTAB(L)[64 + lane] = P.t1; L_CTAB(L)[128 + lane] = P.t2; L_CTAB(L)[192 + lane] = P.t3;
        }
    }
    PCAMV_WAVE_SYNC();
}

/* ---------------------------------------------------------------- intra prediction SATD */
/* 16x16: lane = mode * 16 + block (raster); modes V, H, DC (the variant the neighbours allow), P.  avail: bit 0 left, bit 1 top */
__device__ __forceinline__ void prim_intra16_satd(const FrameDev &F, MBLocal *L, int avail_)
{
    const int avail = rfl(avail_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    const uint8_t *top = L->ib_top[0] + 4, *left = L->ib_left[0];
    /* sums for DC (lanes 0..15 top, 16..31 left) and the plane gradients (lanes 32..39 H, 40..47 V) */
    int v = 0;
    if (lane < 16) v = top[lane];
    else if (lane < 32) v = left[lane - 16];
    else if (lane < 48) {
        const int i = (lane & 7) + 1, isv = (lane >> 3) & 1;
        const int hi = isv ? left[7 + i] : top[7 + i], lo = (7 - i < 0) ? top[-1] : isv ? left[7 - i] : top[7 - i];
        v = i * (hi - lo);
    }
    const int s16 = group_sum(v, 16), s8 = group_sum(v, 8);
    const int st = __builtin_amdgcn_readlane(s16, 0), sl = __builtin_amdgcn_readlane(s16, 16);
    const int H = __builtin_amdgcn_readlane(s8, 32), V = __builtin_amdgcn_readlane(s8, 40);
    const int dc = avail == 3 ? (st + sl + 16) >> 5 : avail == 1 ? (sl + 8) >> 4 : avail == 2 ? (st + 8) >> 4 : 128;
    const int mode = lane >> 4, blk = lane & 15, px = 4 * (blk & 3), py = 4 * (blk >> 2);
    uint32_t r[4], e[4], ec[8];
    if (mode == 0) { const uint32_t t4 = lds4(top + px); r[0] = r[1] = r[2] = r[3] = t4; }
    else if (mode == 1) { r[0] = rep4(left[py]); r[1] = rep4(left[py + 1]); r[2] = rep4(left[py + 2]); r[3] = rep4(left[py + 3]); }
    else if (mode == 2) { r[0] = r[1] = r[2] = r[3] = rep4(dc); }
    else {
        const int a = 16 * (left[15] + top[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t o = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) o |= clamp_u8((a + mul24s(b, px + x - 7) + mul24s(c, py + y - 7) + 16) >> 5) << (8 * x);
            r[y] = o;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = lds4(L->fenc + (py + k) * 16 + px);
    int cost;
    if (F.subme > 1) { pk_cols(e, ec); cost = satd4x4_half(ec, r); }
    else { uint32_t sa = 0; for (int k = 0; k < 4; k++) sa = __builtin_amdgcn_sad_u8(e[k], r[k], sa); cost = (int)sa; }
    cost = group_sum(cost, 16);
    const int need = mode == 0 ? 2 : mode == 1 ? 1 : mode == 3 ? 3 : 0;
    if (blk == 0) L->ccost[mode] = (avail & need) == need ? cost : PCAMV_COST_MAX;
    PCAMV_WAVE_SYNC();
}
/* chroma 8x8: lane = mode * 8 + plane * 4 + block; modes DC (variant), H, V, P; the cost of a mode is over both planes */
__device__ __forceinline__ void prim_intra8c_satd(const FrameDev &F, MBLocal *L, int avail_)
{
    const int avail = rfl(avail_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    int cost = 0;
    const int mode = (lane >> 3) & 3, plane = (lane >> 2) & 1, blk = lane & 3, px = 4 * (blk & 1), py = 4 * (blk >> 1);
    if (lane < 32) {
        const uint8_t *top = L->ib_top[1 + plane] + 4, *left = L->ib_left[1 + plane];
        uint32_t r[4], e[4], ec[8];
        if (mode == 2) { const uint32_t t4 = lds4(top + px); r[0] = r[1] = r[2] = r[3] = t4; }
        else if (mode == 1) { r[0] = rep4(left[py]); r[1] = rep4(left[py + 1]); r[2] = rep4(left[py + 2]); r[3] = rep4(left[py + 3]); }
        else if (mode == 0) {
            const int s0 = bsum4(lds4(top)), s1 = bsum4(lds4(top + 4));
            const int s2 = left[0] + left[1] + left[2] + left[3], s3 = left[4] + left[5] + left[6] + left[7];
            int dc;
            if (avail == 3) dc = blk == 0 ? (s0 + s2 + 4) >> 3 : blk == 1 ? (s1 + 2) >> 2 : blk == 2 ? (s3 + 2) >> 2 : (s1 + s3 + 4) >> 3;
            else if (avail == 1) dc = blk < 2 ? (s2 + 2) >> 2 : (s3 + 2) >> 2;
            else if (avail == 2) dc = (blk & 1) ? (s1 + 2) >> 2 : (s0 + 2) >> 2;
            else dc = 128;
            r[0] = r[1] = r[2] = r[3] = rep4(dc);
        } else {
            int H = 0, V = 0;
#pragma unroll
            for (int i = 1; i <= 4; i++) { H += i * (top[3 + i] - top[3 - i]); V += i * ((int)left[3 + i] - (3 - i < 0 ? (int)top[-1] : (int)left[3 - i])); }
            const int a = 16 * (left[7] + top[7]), b = (17 * H + 16) >> 5, c = (17 * V + 16) >> 5;
#pragma unroll
            for (int y = 0; y < 4; y++) {
                uint32_t o = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) o |= clamp_u8((a + mul24s(b, px + x - 3) + mul24s(c, py + y - 3) + 16) >> 5) << (8 * x);
                r[y] = o;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = lds4(L->fenc + 256 + (py + k) * 16 + plane * 8 + px);
        if (F.subme > 1) { pk_cols(e, ec); cost = satd4x4_half(ec, r); }
        else { uint32_t sa = 0; for (int k = 0; k < 4; k++) sa = __builtin_amdgcn_sad_u8(e[k], r[k], sa); cost = (int)sa; }
    }
    cost = group_sum(cost, 8);
    const int need = mode == 1 ? 1 : mode == 2 ? 2 : mode == 3 ? 3 : 0;
    if (lane < 32 && (lane & 7) == 0) L->ccost[mode] = (avail & need) == need ? cost : PCAMV_COST_MAX;
    PCAMV_WAVE_SYNC();
}
/* the 4x4 analysis works in a picture of its own (L_IFD: 17 rows of 32, row -1 / column -1 = the neighbours) */
__device__ __forceinline__ void prim_intra4_init(MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    if (lane < 25) IFD(L, lane - 1, -1) = L->ib_top[0][3 + lane];
    else if (lane < 41) IFD(L, -1, lane - 25) = L->ib_left[0][lane - 25];
    PCAMV_WAVE_SYNC();
}
#define PF1(a, b) (((a) + (b) + 1) >> 1)
#define PF2(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
/* H.264 8.3.1.2 as common/predict.c:345-487 computes it: the 4x4 prediction of `mode` at (bx, by) of the analysis picture */
__device__ __forceinline__ void ipred4_rows(const MBLocal *L, int bx, int by, int mode, uint32_t r[4])
{
    const uint8_t *base = L_IFD(L) + by * 32 + bx + 4;           /* row by - 1 */
    const uint32_t t0 = lds4(base), t1 = lds4(base + 4);
    int t[8], l[4], e[9];
#pragma unroll
    for (int i = 0; i < 4; i++) { t[i] = (int)(t0 >> (8 * i) & 255); t[4 + i] = (int)(t1 >> (8 * i) & 255); l[i] = base[(i + 1) * 32 - 1]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { e[3 - i] = l[i]; e[5 + i] = t[i]; }
    e[4] = base[-1];
    int p[4][4];
    switch (mode) {
    case I4_V: r[0] = r[1] = r[2] = r[3] = t0; return;
    case I4_H: r[0] = rep4(l[0]); r[1] = rep4(l[1]); r[2] = rep4(l[2]); r[3] = rep4(l[3]); return;
    case I4_DC: r[0] = r[1] = r[2] = r[3] = rep4((l[0] + l[1] + l[2] + l[3] + t[0] + t[1] + t[2] + t[3] + 4) >> 3); return;
    case I4_DC_LEFT: r[0] = r[1] = r[2] = r[3] = rep4((l[0] + l[1] + l[2] + l[3] + 2) >> 2); return;
    case I4_DC_TOP: r[0] = r[1] = r[2] = r[3] = rep4((t[0] + t[1] + t[2] + t[3] + 2) >> 2); return;
    case I4_DC_128: r[0] = r[1] = r[2] = r[3] = 0x80808080u; return;
    case I4_DDL:
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) p[y][x] = (x == 3 && y == 3) ? PF2(t[6], t[7], t[7]) : PF2(t[x + y], t[x + y + 1], t[x + y + 2]);
        break;
    case I4_DDR:
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) p[y][x] = PF2(e[3 + x - y], e[4 + x - y], e[5 + x - y]);
        break;
    case I4_VR:
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int z = 2 * x - y, c = 4 + x - (y >> 1);
                p[y][x] = z < -1 ? PF2(e[4 - y], e[5 - y], e[6 - y]) : (z & 1) ? PF2(e[c - 1], e[c], e[c + 1]) : PF1(e[c], e[c + 1]);
            }
        break;
    case I4_HD:
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int z = 2 * y - x, q = y - (x >> 1);
                p[y][x] = z < -1 ? PF2(e[2 + x], e[3 + x], e[4 + x]) : (z & 1) ? PF2(e[5 - q], e[4 - q], e[3 - q]) : PF1(e[4 - q], e[3 - q]);
            }
        break;
    case I4_VL:
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) { const int k = x + (y >> 1); p[y][x] = (y & 1) ? PF2(t[k], t[k + 1], t[k + 2]) : PF1(t[k], t[k + 1]); }
        break;
    default:    /* I4_HU */
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int z = x + 2 * y, q = y + (x >> 1);
                p[y][x] = z > 5 ? l[3] : z == 5 ? PF2(l[2], l[3], l[3]) : (z & 1) ? PF2(l[q], l[q + 1], l[q + 2]) : PF1(l[q], l[q + 1]);
            }
        break;
    }
#pragma unroll
    for (int y = 0; y < 4; y++) r[y] = (uint32_t)p[y][0] | (uint32_t)p[y][1] << 8 | (uint32_t)p[y][2] << 16 | (uint32_t)p[y][3] << 24;
}
/* SATD of the n listed modes (L->slots) of block idx, one mode per lane -> L->ccost[i] */
__device__ __forceinline__ void prim_intra4_costs(const FrameDev &F, MBLocal *L, int idx_, int n_, int emulate_)
{
    const int idx = rfl(idx_), n = rfl(n_), emulate = rfl(emulate_);
    const int lane = LANE();
    const int bx = 4 * blk_x_of(idx), by = 4 * blk_y_of(idx);
    PCAMV_WAVE_SYNC();
    if (emulate) { *(uint32_t *)&IFD(L, bx + 4, by - 1) = rep4(IFD(L, bx + 3, by - 1)); PCAMV_WAVE_SYNC(); }     /* missing top right samples (analyse.c:812-814) */
    if (lane < n) {
        uint32_t r[4], e[4], ec[8];
        ipred4_rows(L, bx, by, L->slots[lane], r);
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = lds4(L->fenc + (by + k) * 16 + bx);
        int cost;
        if (F.subme > 1) { pk_cols(e, ec); cost = satd4x4_half(ec, r); }
        else { uint32_t sa = 0; for (int k = 0; k < 4; k++) sa = __builtin_amdgcn_sad_u8(e[k], r[k], sa); cost = (int)sa; }
        L->ccost[lane] = cost;
    }
    PCAMV_WAVE_SYNC();
}
/* x264_mb_encode_i4x4 (encoder/macroblock.c:116-148): the block's reconstruction replaces it in the analysis picture */
__device__ __forceinline__ void prim_intra4_encode(const FrameDev &F, MBLocal *L, int idx_, int mode_)
{
    const int idx = rfl(idx_), mode = rfl(mode_);
    const int bx = 4 * blk_x_of(idx), by = 4 * blk_y_of(idx);
    PCAMV_WAVE_SYNC();
    uint32_t r[4];
    ipred4_rows(L, bx, by, mode, r);
    int t[4][4], d[16];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t e = lds4(L->fenc + (by + y) * 16 + bx), p = r[y];
        const int d0 = (int)(e & 255) - (int)(p & 255), d1 = (int)((e >> 8) & 255) - (int)((p >> 8) & 255);
        const int d2 = (int)((e >> 16) & 255) - (int)((p >> 16) & 255), d3 = (int)(e >> 24) - (int)(p >> 24);
        const int s03 = d0 + d3, s12 = d1 + d2, d03 = d0 - d3, d12 = d1 - d2;
        t[0][y] = s03 + s12; t[1][y] = 2 * d03 + d12; t[2][y] = s03 - s12; t[3][y] = d03 - 2 * d12;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
        d[i * 4 + 0] = s03 + s12; d[i * 4 + 1] = 2 * d03 + d12; d[i * 4 + 2] = s03 - s12; d[i * 4 + 3] = d03 - 2 * d12;
    }
    int nz = 0;
    const int qbits = F.qp / 6 - 4;
    int16_t c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int cls = (i & 1) + ((i >> 2) & 1);
        const int mf = cls == 0 ? F.q_mf_i[0] : cls == 1 ? F.q_mf_i[1] : F.q_mf_i[2], bias = cls == 0 ? F.q_bias_i[0] : cls == 1 ? F.q_bias_i[1] : F.q_bias_i[2];
        const int dq = cls == 0 ? F.dq_mf[0] : cls == 1 ? F.dq_mf[1] : F.dq_mf[2];
        int v = d[i];
        const int qa = (int)(mul24u((uint32_t)(bias + iabs(v)), (uint32_t)mf) >> 16);      /* branch-free form, see quant_score_dequant */
        v = v < 0 ? -qa : qa;
        nz |= v;
        c[i] = qbits >= 0 ? (int16_t)(mul24s(v, dq) << qbits) : (int16_t)((mul24s(v, dq) + (1 << (-qbits - 1))) >> (-qbits));
    }
    if (nz) {
        /* add4x4_idct (dct.c:174-216) on top of the prediction rows */
        int16_t tt[4][4], rr[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int s02 = c[0 * 4 + i] + c[2 * 4 + i], d02 = c[0 * 4 + i] - c[2 * 4 + i];
            const int s13 = c[1 * 4 + i] + (c[3 * 4 + i] >> 1), d13 = (c[1 * 4 + i] >> 1) - c[3 * 4 + i];
            tt[i][0] = (int16_t)(s02 + s13); tt[i][1] = (int16_t)(d02 + d13); tt[i][2] = (int16_t)(d02 - d13); tt[i][3] = (int16_t)(s02 - s13);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int s02 = tt[0][i] + tt[2][i], d02 = tt[0][i] - tt[2][i];
            const int s13 = tt[1][i] + (tt[3][i] >> 1), d13 = (tt[1][i] >> 1) - tt[3][i];
            rr[0][i] = (int16_t)((s02 + s13 + 32) >> 6); rr[1][i] = (int16_t)((d02 + d13 + 32) >> 6);
            rr[2][i] = (int16_t)((d02 - d13 + 32) >> 6); rr[3][i] = (int16_t)((s02 - s13 + 32) >> 6);
        }
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t o = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((r[y] >> (8 * x)) & 255) + rr[y][x], 0, 255) << (8 * x);
            r[y] = o;
        }
    }
    PCAMV_WAVE_SYNC();
    if (LANE() == 0) {
#pragma unroll
        for (int y = 0; y < 4; y++) *(uint32_t *)&IFD(L, bx, by + y) = r[y];
        L->nzc[scan8_of(idx)] = (uint8_t)(nz != 0);     /* encoder/macroblock.c:135: stays in the cache (the sub-partition RD trials read it) */
    }
    PCAMV_WAVE_SYNC();
}

/* ---------------------------------------------------------------- psy-RD complexity + SSD */
/* 4x4 Hadamard coefficients of the pixel rows (against zero) */
__device__ __forceinline__ void had4x4_coefs(const uint32_t r[4], int t[16])
{
    int h[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const int p0 = (int)(r[y] & 255), p1 = (int)((r[y] >> 8) & 255), p2 = (int)((r[y] >> 16) & 255), p3 = (int)(r[y] >> 24);
        const int s01 = p0 + p1, d01 = p0 - p1, s23 = p2 + p3, d23 = p2 - p3;
        h[y][0] = s01 + s23; h[y][1] = d01 + d23; h[y][2] = s01 - s23; h[y][3] = d01 - d23;
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int s01 = h[0][x] + h[1][x], d01 = h[0][x] - h[1][x], s23 = h[2][x] + h[3][x], d23 = h[2][x] - h[3][x];
        t[x] = s01 + s23; t[4 + x] = d01 + d23; t[8 + x] = s01 - s23; t[12 + x] = d01 - d23;
    }
}
/* lane < 16 (x264 block order: quads of lanes are 8x8 blocks): sum of |4x4 coefficients| of its block, its pixel sum, and
 * its share of the sum of |8x8 coefficients| of its 8x8 (the 8x8 Hadamard = 2x2 Hadamard across the four 4x4 transforms) */
__device__ __forceinline__ void had_lane_sums(const uint8_t *buf, int lane, int *s4, int *dc, int *s8)
{
    uint32_t r[4]; int t[16];
    const int px = 4 * blk_x_of(lane & 15), py = 4 * blk_y_of(lane & 15);
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = lds4(buf + (py + k) * 16 + px);
    had4x4_coefs(r, t);
    int a4 = 0, a8 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { a4 += iabs(t[k]); a8 += iabs(quad_had2x2(t[k], lane & 3)); }
    *s4 = a4; *dc = t[0]; *s8 = a8;
}
/* The same energies with the whole wave at work: lane = 4 * block (x264 order) + row of the block, so a block is a quad of
 * lanes and an 8x8 a row of 16.  A lane transforms its row of four pixels (packed 16-bit: (p0 + p1, p2 + p3), (p0 - p1, p2 - p3);
 * the last butterfly of the horizontal transform is never made: sum |x + y| + |x - y| = 2 max(|x|, |y|)), the vertical transform
 * runs across the quad and the 2x2 of the 8x8 across the row of 16 (butterflies as partner + sign * own: the sign of a lane's
 * whole vector is free, every sum taken is of magnitudes).  Returns the lane's shares of half the sum of |4x4 coefficients|,
 * half the sum of |8x8 coefficients| and the pixel sum of its row.  Values stay below 8160 * 2: 16-bit is exact. */
__device__ __forceinline__ v2s had_bfly(v2s x, int partner, bool minus)
{
    const v2s one = {1, 1}, mone = {-1, -1};
    return x * (minus ? mone : one) + as_v2s((uint32_t)partner);
}
__device__ __forceinline__ int had_pairmax(v2s s, v2s d)
{
    const v2s z = {0, 0};
    const v2s as = __builtin_elementwise_max(s, z - s), ad = __builtin_elementwise_max(d, z - d);
    const v2s m = __builtin_elementwise_max(as, __builtin_shufflevector(as, as, 1, 0)) + __builtin_elementwise_max(ad, __builtin_shufflevector(ad, ad, 1, 0));
    return (int)(as_u32(m) & 0xffffu);
}
__device__ __forceinline__ void had_row_shares(const uint8_t *buf, int lane, int *h4, int *h8, int *dc)
{
    const int b = lane >> 2;
    const uint32_t row = lds4(buf + (4 * blk_y_of(b) + (lane & 3)) * 16 + 4 * blk_x_of(b));
    const v2s A = as_v2s(__builtin_amdgcn_perm(0u, row, 0x0c020c00u)), B = as_v2s(__builtin_amdgcn_perm(0u, row, 0x0c030c01u));      /* (p0, p2), (p1, p3) */
    v2s S = A + B, D = A - B;
    S = had_bfly(S, dpp_qp1((int)as_u32(S)), lane & 1); D = had_bfly(D, dpp_qp1((int)as_u32(D)), lane & 1);
    S = had_bfly(S, dpp_qp2((int)as_u32(S)), lane & 2); D = had_bfly(D, dpp_qp2((int)as_u32(D)), lane & 2);
    *h4 = had_pairmax(S, D);
    *dc = (int)__builtin_amdgcn_sad_u8(row, 0u, 0u);
    /* lane ^ 4, lane ^ 8 inside the row of 16: two row shifts, each written to the banks (quads) it is right for */
    {
        int ps = __builtin_amdgcn_update_dpp(0, (int)as_u32(S), 0x104, 0xf, 0x5, false); ps = __builtin_amdgcn_update_dpp(ps, (int)as_u32(S), 0x114, 0xf, 0xa, false);
        int pd = __builtin_amdgcn_update_dpp(0, (int)as_u32(D), 0x104, 0xf, 0x5, false); pd = __builtin_amdgcn_update_dpp(pd, (int)as_u32(D), 0x114, 0xf, 0xa, false);
        S = had_bfly(S, ps, lane & 4); D = had_bfly(D, pd, lane & 4);
    }
    {
        int ps = __builtin_amdgcn_update_dpp(0, (int)as_u32(S), 0x108, 0xf, 0x3, false); ps = __builtin_amdgcn_update_dpp(ps, (int)as_u32(S), 0x118, 0xf, 0xc, false);
        int pd = __builtin_amdgcn_update_dpp(0, (int)as_u32(D), 0x108, 0xf, 0x3, false); pd = __builtin_amdgcn_update_dpp(pd, (int)as_u32(D), 0x118, 0xf, 0xc, false);
        S = had_bfly(S, ps, lane & 8); D = had_bfly(D, pd, lane & 8);
    }
    *h8 = had_pairmax(S, D);
}
__device__ __forceinline__ void prim_fenc_complexity(const FrameDev &F, MBLocal *L)      /* x264_mb_cache_fenc_satd */
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    int satd = 0, sa8d = 0;
    if (F.psy_rd) {
        int h4, h8, dc;
        had_row_shares(L->fenc, lane, &h4, &h8, &dc);
        /* per 4x4: satd - (dc >> 1) with satd = sum |coefficients| >> 1; per 8x8: ((sum |coefficients| + 2) >> 2) - (dc >> 2) */
        int b4 = h4 + dpp_qp1(h4); b4 += dpp_qp2(b4);
        int d4 = dc + dpp_qp1(dc); d4 += dpp_qp2(d4);
        const int b8 = group_sum(h8, 16), d8 = group_sum(dc, 16);
        int a = (lane & 3) == 0 ? b4 - (d4 >> 1) : 0;
        int b = (lane & 15) == 0 ? ((2 * b8 + 2) >> 2) - (d8 >> 2) : 0;
        satd = wave_sum_all(a); sa8d = wave_sum_all(b);
    }
    if (lane == 0) { L->fenc_satd_sum = satd; L->fenc_sa8d_sum = sa8d; }
    PCAMV_WAVE_SYNC();
}
/* x264_pixel_hadamard_ac_16x16 (common/pixel.c:306-358) of the 16x16 block in buf (stride 16): AC energy of the 4x4 transforms
 * (low word of the reference's result) and of the 8x8 transforms (high word) */
__device__ __forceinline__ void prim_hadamard_ac16(const uint8_t *buf, int lane, int *sum4, int *sum8)
{
    int h4, h8, dc;
    had_row_shares(buf, lane, &h4, &h8, &dc);
    const int t4 = wave_sum_all(h4), t8 = wave_sum_all(h8), td = wave_sum_all(dc);
    *sum4 = (2 * t4 - td) >> 1;
    *sum8 = (2 * t8 - td) >> 2;
}
/* ssd_mb (rdo.c:106-137) of the reconstruction in L->pred: SSD of luma + both chroma planes, plus for luma the psy term
 * |AC energy (4x4) difference| + |AC energy (8x8) difference| (the hadamard_ac branch: PIXEL_16x16 <= PIXEL_8x8).
 * A row of four pixels per lane (64 luma rows, then the 32 chroma rows on half the wave): sum (e - p)^2 = e.e + p.p - 2 e.p,
 * three byte dot products */
__device__ __forceinline__ int prim_ssd_mb(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    const int b = lane >> 2, rr = lane & 3;
    int v;
    {
        const int o = (4 * blk_y_of(b) + rr) * 16 + 4 * blk_x_of(b);
        const uint32_t e = lds4(L->fenc + o), p = lds4(L->pred + o);
        v = (int)__builtin_amdgcn_udot4(p, p, __builtin_amdgcn_udot4(e, e, 0u, false), false) - 2 * (int)__builtin_amdgcn_udot4(e, p, 0u, false);
    }
    if (lane < 32) {
        const int o = (16 + (b & 2) * 2 + rr) * 16 + (b >> 2) * 8 + (b & 1) * 4;          /* b = 4 * plane + block of the plane */
        const uint32_t e = lds4(L->fenc + o), p = lds4(L->pred + o);
        v += (int)__builtin_amdgcn_udot4(p, p, __builtin_amdgcn_udot4(e, e, 0u, false), false) - 2 * (int)__builtin_amdgcn_udot4(e, p, 0u, false);
    }
    int ssd = wave_sum_all(v);
    if (F.psy_rd) {
        int sum4, sum8;
        prim_hadamard_ac16(L->pred, lane, &sum4, &sum8);
        const int satd = (iabs(sum4 - L->fenc_satd_sum) + iabs(sum8 - L->fenc_sa8d_sum)) >> 1;
        ssd += (satd * F.psy_rd * F.lambda + 128) >> 8;
    }
    PCAMV_WAVE_SYNC();
    return ssd;
}

/* ---------------------------------------------------------------- CABAC size / context walk of a macroblock */
/* The size of a decision depends on the state of its context only, a context's state on the decisions made on THAT context only,
 * and which decisions a macroblock makes (context, bin, order) on its syntax elements only -- never on a state.  So the walk has
 * two halves.  The wave-uniform control code that follows the syntax (macroblock header; the level chains of the residual) does
 * not look anything up: a decision is appended to the private queue of the lane that owns its context (64 bins in two
 * registers + a count: a compare, a shift, an add).  Then every lane walks its own queue -- state -> (bits << 8 | next state)
 * out of the table in LDS, one gather per step for ALL contexts at once -- so the length of the chain is the longest queue (the
 * mvd contexts' 9 at most in a header; the level contexts' few dozen in a busy macroblock) instead of the number of decisions
 * (30 .. 200), each of which used to be a dependent round of v_readlane / table select / write-back on the scalar side
 * (15 k cycles per header, 28 k per residual of a lone wave's macroblock).  Header contexts 0..87 (mb_skip, mb_type, sub_mb_type,
 * mvd, mb_qp_delta, coded_block_pattern): context c in lane c & 63 of register c >> 6.  The slice's states (LDS) are read at
 * the beginning; the states the walk ends in go to the slice's (committing walk) or to the trial copy. */
/* (round 3: ONE state / queue register pair per lane.  A P slice's macroblock header touches contexts 11..23, 40..53, 60 and 73..84
 * only; lanes 24..35 -- whose own contexts, B-slice macroblock types, it never touches -- hold 73..84, so a header is one queue
 * walk, not two, and a decision needs no choice of register.) */
struct CabWalk { int s, n, tot, bits, vbits; unsigned long long q; };
__device__ __forceinline__ void cabq_push(unsigned long long &q, int &n, int owner, int bin)
{
    const bool me = LANE() == owner;
    q |= me ? (unsigned long long)(unsigned)bin << n : 0ull;
    n += me ? 1 : 0;
}
/* `ones` 1-bins followed, if `zero`, by one 0-bin: the unary prefix of a level / of an MV difference */
__device__ __forceinline__ void cabq_push_run(unsigned long long &q, int &n, int owner, int ones, int zero)
{
    const bool me = LANE() == owner;
    q |= me ? ((1ull << ones) - 1ull) << n : 0ull;
    n += me ? ones + zero : 0;
}
/* every lane walks its queue; returns the bits (8.8 fixed point) of this lane's decisions */
__device__ __forceinline__ int cabq_resolve(const uint32_t *T, int &st, unsigned long long &q, int &n)
{
    int bits = 0;
    while (__builtin_amdgcn_ballot_w64(n > 0)) {
        if (n > 0) {
            const uint32_t w = T[2 * st + (int)(q & 1ull)];
            bits += (int)(w >> 8); st = (int)(w & 255u);
            q >>= 1; n--;
        }
    }
    q = 0;
    return bits;
}
__device__ __forceinline__ int cab_hdr_ctx(int lane) { return lane >= 24 && lane < 36 ? lane + 49 : lane; }       /* the context a lane owns during a header */
__device__ __forceinline__ int cab_hdr_lane(int ctx) { return ctx < 64 ? ctx : ctx - 49; }
/* trial: a size trial -- its end states go to the trial copy (which starts as a copy of the slice's), never to the slice's */
__device__ __forceinline__ void prim_cab_begin(MBLocal *L, CabWalk &C, int trial_)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    const uint8_t *S = L_CAB(L, 0);
    if (rfl(trial_)) {
        const uint32_t *s = (const uint32_t *)S; uint32_t *d = (uint32_t *)L_CABT(L);
        d[lane] = s[lane]; if (lane < PCAMV_CAB_USED / 4 - 64) d[64 + lane] = s[64 + lane];
    }
    C.s = S[cab_hdr_ctx(lane)]; C.bits = 0; C.vbits = 0;
    C.q = 0; C.n = 0; C.tot = 0;
}
__device__ __forceinline__ void prim_cab_flush(MBLocal *L, CabWalk &C)
{
    C.vbits += cabq_resolve(L_CTAB(L), C.s, C.q, C.n);
    C.tot = 0;
}
__device__ __forceinline__ void prim_cb_dec(MBLocal *L, CabWalk &C, int ctx_, int b_)
{
    const int ctx = rfl(ctx_), b = rfl(b_);
    /* (no queue may pass 64 bins: the decisions since the last walk are counted on the scalar side; 16 MV differences of a macroblock
     * in 4x4 partitions can put more than that on one context) */
    if (rfl(C.tot) >= 48) prim_cab_flush(L, C);
    cabq_push(C.q, C.n, cab_hdr_lane(ctx), b);
    C.tot++;
}
/* `ones` 1-bins and, if zero_, a 0-bin on one context (ones + zero_ <= 16) */
__device__ __forceinline__ void prim_cb_run(MBLocal *L, CabWalk &C, int ctx_, int ones_, int zero_)
{
    const int ctx = rfl(ctx_), ones = rfl(ones_), zero = rfl(zero_);
    if (rfl(C.tot) >= 48) prim_cab_flush(L, C);
    cabq_push_run(C.q, C.n, cab_hdr_lane(ctx), ones, zero);
    C.tot += ones + zero;
}
__device__ __forceinline__ void prim_cb_bypass(CabWalk &C, int f8) { C.bits += rfl(f8); }
__device__ __forceinline__ int prim_cab_end(MBLocal *L, CabWalk &C, int commit_)
{
    const int lane = LANE();
    const uint32_t *T = L_CTAB(L);
    C.vbits += cabq_resolve(T, C.s, C.q, C.n);
    {
        uint8_t *S = rfl(commit_) ? L_CAB(L, 0) : L_CABT(L);
        S[cab_hdr_ctx(lane)] = (uint8_t)C.s;
    }
    const int total = rfl(C.bits) + wave_sum_all(C.vbits);
    PCAMV_WAVE_SYNC();
    return total;
}
/* coded_block_flag + significance map + levels of every coded block (encoder/cabac.c:540-667, 1000-1018), three steps:
 *  1. one block per lane (0..15 luma 4x4, 16..23 chroma AC, 24 / 25 chroma DC): is it coded (coded_block_pattern), its
 *     coded_block_flag and that flag's context increment (left / top neighbours), its levels as bit masks (non-zero, above 1)
 *     and 4-bit magnitudes -- one round of LDS reads for the whole macroblock;
 *  2. the coded_block_flag chains, one context per lane (3 categories x 4 increments), each lane walking the blocks that use it;
 *  3. per category, the blocks with levels in coding order: lane i owns significant_coeff_flag[i] and last_significant_
 *     coeff_flag[i] (one decision each per block, all positions at once), lane k < 10 owns coeff_abs_level_minus1 context k
 *     and takes the level chain's decisions as they come (non-zero levels only, from the last one down).
 * The per-block values of step 1 reach the wave-uniform control code of steps 2 / 3 through ballots and v_readlane. */
/* The walk is a REAL function (PCAMV_RESIDUAL_CALL, the RD kernels): inlined into the decision code at 128 VGPRs its step 1
 * alone carried 300 scratch reloads of values that were live around it and took 17 % of the search's wave time under load
 * (1.6 M of 9.7 M cycles per macroblock at 4096 chains, against 2 % for the build without spills); as a callee it has the
 * register file to itself, and what it exchanges with the caller is the table (4 registers) in, two bit counts out. */
struct CabBits { int bits, vbits; };
#ifdef PCAMV_RESIDUAL_CALL
#define PCAMV_RESIDUAL_FN static __device__ __noinline__
#else
#define PCAMV_RESIDUAL_FN __device__ __forceinline__
#endif
PCAMV_RESIDUAL_FN CabBits cab_residual_walk(MBLocal *L, int commit_, int part_)
{
    CabBits out = {0, 0};
    const int commit = rfl(commit_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    const int cbp_luma = rfl(L->cbp_luma), cbp_chroma = rfl(L->cbp_chroma);
    if (!(cbp_luma | cbp_chroma)) return out;
    uint8_t *S = L_CAB(L, 0);
    uint8_t *D = commit ? S : L_CABT(L);                 /* where the states the walk ends in go */
    const uint32_t *T = L_CTAB(L);
    int bits = 0, sbits = 0;            /* per-lane bits (map decisions, coded_block_flag chains) / wave-uniform bits (level chains) */
    const unsigned long long t_1 = PROF_T();
    /* ---- 1 */
    /* part >= 0: x264_partition_size_cabac of 8x8 `part` (encoder/cabac.c:1058-1074): its luma blocks when it is coded, its two chroma AC blocks */
    const int part = rfl(part_);
    const bool coded = part >= 0 ? (lane < 16 ? (lane >> 2) == part && ((cbp_luma >> part) & 1) != 0 : lane < 24 ? ((lane - 16) & 3) == part : false)
                     : lane < 16 ? ((cbp_luma >> (lane >> 2)) & 1) != 0 : lane < 24 ? (cbp_chroma & 2) != 0 : lane < 26 ? (cbp_chroma & 3) != 0 : false;
    const int idx = lane < 24 ? lane : lane < 26 ? 25 + (lane - 24) : 0;
    const int count = lane < 16 ? 16 : lane < 24 ? 15 : 4;
    int flag = 0, inc = 0;
    unsigned nzm = 0, gt1 = 0, nib0 = 0, nib1 = 0;
    if (coded) {
        const int p8 = scan8_all_of(idx);
        flag = L->nzc[p8] != 0;
        if (lane >= 24) {
            const int cl = L->cbp_left, ct = L->cbp_top, k = lane - 24;
            inc = (cl != -1 ? (cl >> (9 + k)) & 1 : 0) + 2 * (ct != -1 ? (ct >> (9 + k)) & 1 : 0);
        } else inc = ((L->nzc[p8 - 1] & 0x7f) != 0) + 2 * ((L->nzc[p8 - 8] & 0x7f) != 0);
        if (flag) {
            const uint32_t *w = lane < 24 ? (const uint32_t *)L->coef[lane] : (const uint32_t *)L->cdc[lane - 24];
            const int nw = lane < 24 ? 8 : 2, sh = (lane >= 16 && lane < 24) ? 1 : 0;       /* chroma AC: scan position i is raw[i + 1] */
            unsigned nzr = 0, g1r = 0;
            unsigned long long nib = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t v = j < nw ? w[j] : 0u;
                const unsigned a0 = (unsigned)imin(iabs((int)(int16_t)(v & 0xffffu)), 15), a1 = (unsigned)imin(iabs((int)v >> 16), 15);
                nzr |= (unsigned)(a0 != 0) << (2 * j) | (unsigned)(a1 != 0) << (2 * j + 1);
                g1r |= (unsigned)(a0 > 1) << (2 * j) | (unsigned)(a1 > 1) << (2 * j + 1);
                nib |= (unsigned long long)(a0 | a1 << 4) << (8 * j);
            }
            nzm = nzr >> sh; gt1 = g1r >> sh; nib >>= 4 * sh;
            nib0 = (unsigned)nib; nib1 = (unsigned)(nib >> 32);
        }
    }
    const unsigned codedm = (unsigned)__ballot(coded), flagm = (unsigned)__ballot(flag != 0 && nzm != 0);
    const unsigned inc_lo = (unsigned)__ballot(coded && (inc & 1)), inc_hi = (unsigned)__ballot(coded && (inc & 2));
    PROF_ADD(24, t_1);
    PROF_CNT(27, 1); PROF_CNT(28, __builtin_popcount(flagm)); PROF_CNT(29, __builtin_popcount(codedm));
    const unsigned long long t_2 = PROF_T();
    /* ---- 2: lane = 4 * category + increment (categories luma 4x4, chroma DC, chroma AC = block bits 0..15, 24..25, 16..23) */
    if (lane < 12) {
        const int cat = lane >> 2, k = lane & 3;
        const unsigned catm = cat == 0 ? 0xffffu : cat == 1 ? 0x3000000u : 0xff0000u;
        const unsigned same = ~(inc_lo ^ (0u - (unsigned)(k & 1))) & ~(inc_hi ^ (0u - (unsigned)(k >> 1)));
        int st = S[85 + 4 * (2 + cat) + k];
        for (unsigned m = codedm & catm & same; m; m &= m - 1) {
            const int b = __builtin_ctz(m);
            const uint32_t w = T[2 * st + (int)((flagm >> b) & 1u)];
            bits += (int)(w >> 8); st = (int)(w & 255u);
        }
        D[85 + 4 * (2 + cat) + k] = (uint8_t)st;
    }
    PROF_ADD(25, t_2);
    const unsigned long long t_3 = PROF_T();
#ifdef PCAMV_RESIDUAL_V1
    /* ---- 3 */
    for (int pass = 0; pass < 3; pass++) {
        const unsigned catm = pass == 0 ? 0xffffu : pass == 1 ? 0x3000000u : 0xff0000u;
        unsigned bm = flagm & catm;
        if (!bm) continue;
        const int cnt = pass == 0 ? 16 : pass == 1 ? 4 : 15;
        const int sig_off = pass == 0 ? 134 : pass == 1 ? 149 : 152, last_off = pass == 0 ? 195 : pass == 1 ? 210 : 213, lvl_off = pass == 0 ? 247 : pass == 1 ? 257 : 266;
        int sigS = lane < cnt - 1 ? S[sig_off + lane] : 0, lastS = lane < cnt - 1 ? S[last_off + lane] : 0, lvlS = lane < 10 ? S[lvl_off + lane] : 0;
        unsigned long long lvlQ = 0; int lvlN = 0, lvl_tot = 0;
        for (; bm; bm &= bm - 1) {
            const int b = __builtin_ctz(bm);
            const unsigned nz = (unsigned)__builtin_amdgcn_readlane((int)nzm, b), g1 = (unsigned)__builtin_amdgcn_readlane((int)gt1, b);
            const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nib0, b), n1 = (unsigned)__builtin_amdgcn_readlane((int)nib1, b);
            const int last = 31 - __builtin_clz(nz);
            if (lane < imin(last + 1, cnt - 1)) {
                const int sb = (int)((nz >> lane) & 1u);
                const uint32_t w1 = T[2 * sigS + sb], w2 = T[2 * lastS + (lane == last)];
                bits += (int)(w1 >> 8); sigS = (int)(w1 & 255u);
                if (sb) { bits += (int)(w2 >> 8); lastS = (int)(w2 & 255u); }
            }
            /* levels from the last non-zero one down: node = min(#(|l| = 1) so far, 3) until a level above 1 was seen, then
             * min(3 + #(|l| > 1), 7); the decisions go to the queues of the lanes that own the level contexts (lane k < 10) */
            int neq1 = 0, ngt1 = 0;
            for (unsigned m = nz; m;) {
                const int i = 31 - __builtin_clz(m);
                m &= ~(1u << i);
                const int node = ngt1 ? imin(3 + ngt1, 7) : imin(neq1, 3);
                const int c1 = node < 4 ? node + 1 : 0, c2 = node < 4 ? 5 : imin(node + 2, 9);
                if (lvl_tot > 64 - 16) { bits += cabq_resolve(T, lvlS, lvlQ, lvlN); lvl_tot = 0; }     /* no queue may pass 64 bins */
                if ((g1 >> i) & 1u) {
                    int a = (int)(((i < 8 ? n0 : n1) >> (4 * (i & 7))) & 15u);
                    if (a == 15) {          /* 15 or more: the exact magnitude (escape suffix) from the block's levels */
                        const int16_t *l = b < 16 ? L->coef[b] : b < 24 ? L->coef[b] + 1 : L->cdc[b - 24];
                        a = rfl(iabs((int)l[i]));
                    }
                    const int am1 = a - 1, prefix = imin(am1, 14);
                    cabq_push(lvlQ, lvlN, c1, 1);
                    cabq_push_run(lvlQ, lvlN, c2, prefix - 1, prefix < 14);
                    if (prefix >= 14) sbits += size_ue_of((unsigned)(am1 - 14)) << 8;
                    lvl_tot += 1 + prefix;
                    ngt1++;
                } else {
                    cabq_push(lvlQ, lvlN, c1, 0);
                    lvl_tot++;
                    neq1++;
                }
            }
            sbits += 256 * __builtin_popcount(nz);       /* signs */
            PROF_CNT(30, __builtin_popcount(nz));
        }
        bits += cabq_resolve(T, lvlS, lvlQ, lvlN);
        if (lane < cnt - 1) { D[sig_off + lane] = (uint8_t)sigS; D[last_off + lane] = (uint8_t)lastS; }
        if (lane < 10) D[lvl_off + lane] = (uint8_t)lvlS;
    }
#else
    /* ---- 3 (round 3): nothing per level or per block on the scalar side any more.
     * 3a. Every block with levels (its own lane) describes, in ten words, what it contributes to each context, in closed form: with
     *     n levels, `last` the highest position, h the highest position of a level above 1 and t the number of levels above h
     *     (= n when there is none above 1), the level chain from the last level down visits node min(r, 3) for its r-th level while
     *     r <= t, then node >= 4 -- so coeff_abs_level_minus1 contexts 1..3 see at most one decision of the block (level r = 0 / 1 / 2:
     *     bin r == t), context 4 the levels 3..min(t, n - 1) (zeros, the last a one when it is level t), context 0 every level below
     *     h (bin = above 1), and the s-th level above 1 puts its unary run on context 5 + min(s, 4); the significance map is the
     *     mask of positions <= last with the non-zero mask as its bins, last_significant the non-zero mask with bit `last` as bins.
     *     Row (LDS): [0] P | counts of contexts 1..3 << 16, [1] nz | their bins << 16, [2] 1 << last, [3] context 0, [4] context 4,
     *     [5..8] contexts 5..8 (bins | count << 26), [9] 0.
     * 3b. One lane per context (significance 15 / 3 / 14, contexts 1..3, last, contexts 0 and 4..9: 40 lanes for luma, 16 for the
     *     chroma DCs beside them, 38 for chroma AC in a second round) appends its bins of block after block to its queue -- one
     *     or two LDS reads and a dozen VALU instructions per block for ALL contexts -- and the queues are walked against the table
     *     once per round (cabq_resolve: the chain is the longest queue).  Only a block's fifth and further levels above 1 (context
     *     9, runs of up to 14 bins each) are pushed from the scalar side as before. */
    uint32_t *ROW = (uint32_t *)L->cxy;                 /* cxy[64] + ccost[192]: idle outside a list evaluation; rows of blocks 0..23 */
    uint32_t *ROWDC = (uint32_t *)L->red;               /* rows of the two chroma DC blocks */
    const bool has_levels = lane < 26 && ((flagm >> lane) & 1u) != 0;       /* (a shift by the lane number wraps at 32) */
    if (has_levels) {
        const unsigned nz = nzm;
        const int n = __builtin_popcount(nz), last = 31 - __builtin_clz(nz);
        const int h = gt1 ? 31 - __builtin_clz(gt1) : -1;
        const int t = gt1 ? __builtin_popcount(nz >> (h + 1)) : n;
        const unsigned c2 = (n >= 2 && t >= 1) ? 1u : 0u, c3 = (n >= 3 && t >= 2) ? 1u : 0u;
        const unsigned b1 = t == 0 ? 1u : 0u, b2 = (c2 && t == 1) ? 1u : 0u, b3 = (c3 && t == 2) ? 1u : 0u;
        const int cnt4 = imax(imin(t, n - 1) - 2, 0);
        const unsigned bits4 = (t < n && t >= 3) ? 1u << (cnt4 - 1) : 0u;
        unsigned bits0 = 0, w5 = 0, w6 = 0, w7 = 0, w8 = 0;
        int r0 = 0, sidx = 0, esc = 0;
        bool first = true;
        for (unsigned mm = gt1 ? nz & ((2u << h) - 1u) : 0u; mm;) {
            const int i = 31 - __builtin_clz(mm);
            mm &= ~(1u << i);
            const unsigned g = (gt1 >> i) & 1u;
            if (!first) { bits0 |= g << r0; r0++; }
            first = false;
            if (g) {
                int a = (int)(((i < 8 ? nib0 : nib1) >> (4 * (i & 7))) & 15u);
                if (a == 15) {          /* 15 or more: the exact magnitude (escape suffix) from the block's levels */
                    const int16_t *l = lane < 16 ? L->coef[lane] : lane < 24 ? L->coef[lane] + 1 : L->cdc[lane - 24];
                    a = iabs((int)l[i]);
                }
                const int am1 = a - 1, prefix = imin(am1, 14), ones = prefix - 1, z = prefix < 14 ? 1 : 0;
                if (prefix >= 14) esc += (2 * (31 - __builtin_clz((unsigned)(am1 - 14) + 1u)) + 1) << 8;       /* size_ue */
                const unsigned w = ((1u << ones) - 1u) | (unsigned)(ones + z) << 26;
                w5 = sidx == 0 ? w : w5; w6 = sidx == 1 ? w : w6; w7 = sidx == 2 ? w : w7; w8 = sidx == 3 ? w : w8;
                sidx++;
            }
        }
        bits += 256 * n + esc;                           /* signs, escape suffixes */
        PROF_CNT(30, n);
        uint32_t *row = lane < 24 ? ROW + 10 * lane : ROWDC + 10 * (lane - 24);
        row[0] = ((2u << last) - 1u) | (1u | c2 << 1 | c3 << 2) << 16;
        row[1] = nz | (b1 | b2 << 1 | b3 << 2) << 16;
        row[2] = 1u << last;
        row[3] = bits0 | (unsigned)r0 << 26;
        row[4] = bits4 | (unsigned)cnt4 << 26;
        row[5] = w5; row[6] = w6; row[7] = w7; row[8] = w8; row[9] = 0u;
    }
    const unsigned manym = (unsigned)__ballot(has_levels && __builtin_popcount(gt1) > 4);
    PCAMV_WAVE_SYNC();
    for (int round = 0; round < 2; round++) {
        const unsigned bm0 = flagm & (round == 0 ? 0x300ffffu : 0xff0000u);
        if (!bm0) continue;
        /* this lane's context: group (category), place in the group */
        const int grp = round == 0 ? (lane < 40 ? 0 : 1) : 2;                         /* 0 luma 4x4, 1 chroma DC, 2 chroma AC */
        const int wl = grp == 1 ? lane - 40 : lane;
        const int ns = grp == 0 ? 15 : grp == 1 ? 3 : 14;
        const int sig_off = grp == 0 ? 134 : grp == 1 ? 149 : 152, last_off = grp == 0 ? 195 : grp == 1 ? 210 : 213, lvl_off = grp == 0 ? 247 : grp == 1 ? 257 : 266;
        const bool k_sig = wl < ns, k_one = !k_sig && wl < ns + 3, k_last = wl >= ns + 3 && wl < 2 * ns + 3, k_wide = wl >= 2 * ns + 3 && wl < 2 * ns + 10;
        const bool live = k_sig || k_one || k_last || k_wide;
        const int j = wl - 2 * ns - 3;                                                   /* wide: 0 = context 0, 1..6 = contexts 4..9 */
        const int colC = k_last ? 1 : k_wide ? 3 + j : 0, colB = k_last ? 2 : k_wide ? 3 + j : 1;
        const int sh = k_sig ? wl : k_one ? 16 + (wl - ns) : k_last ? wl - ns - 3 : 0;
        const int ctx = k_sig ? sig_off + wl : k_one ? lvl_off + 1 + (wl - ns) : k_last ? last_off + (wl - ns - 3) : lvl_off + (j == 0 ? 0 : 3 + j);
        const unsigned mycat = grp == 0 ? 0xffffu : grp == 1 ? 0x3000000u : 0xff0000u;
        const uint32_t *base = grp == 1 ? ROWDC - 240 : ROW;                              /* row of block b = base + 10 b */
        int st = live ? S[ctx] : 0;
        unsigned long long q = 0; int qn = 0;
        unsigned m = bm0;
        int b = __builtin_ctz(m);
        uint32_t cw = live ? base[10 * b + colC] : 0u, bw = live ? base[10 * b + colB] : 0u;
        while (m) {
            m &= m - 1;
            const int bn = m ? __builtin_ctz(m) : b;
            const uint32_t cw_n = live ? base[10 * bn + colC] : 0u, bw_n = live ? base[10 * bn + colB] : 0u;      /* the next block's words while this one's are used */
            if (live && ((mycat >> b) & 1u)) {
                const uint32_t xc = cw >> sh, xb = bw >> sh;
                const int cnt = k_wide ? (int)(xc >> 26) : (int)(xc & 1u);
                const unsigned bins = k_wide ? xc & 0x3ffffffu : xb & 1u;
                q |= (unsigned long long)bins << qn;
                qn += cnt;
            }
            if ((manym >> b) & 1u) {          /* the block's fifth and further levels above 1: context 9, from the scalar side */
                const unsigned g1 = (unsigned)__builtin_amdgcn_readlane((int)gt1, b);
                const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nib0, b), n1 = (unsigned)__builtin_amdgcn_readlane((int)nib1, b);
                int sidx = 0;
                for (unsigned mm = g1; mm; sidx++) {
                    const int i = 31 - __builtin_clz(mm);
                    mm &= ~(1u << i);
                    if (sidx < 4) continue;
                    int a = (int)(((i < 8 ? n0 : n1) >> (4 * (i & 7))) & 15u);
                    if (a == 15) {
                        const int16_t *l = b < 16 ? L->coef[b] : b < 24 ? L->coef[b] + 1 : L->cdc[b - 24];
                        a = rfl(iabs((int)l[i]));
                    }
                    const int prefix = imin(a - 1, 14);
                    if (__builtin_amdgcn_ballot_w64(qn > 64 - 16)) bits += cabq_resolve(T, st, q, qn);
                    cabq_push_run(q, qn, b >= 24 ? 55 : round == 0 ? 39 : 37, prefix - 1, prefix < 14);      /* the lane of the category's context 9 */
                }
            }
            if (__builtin_amdgcn_ballot_w64(qn > 64 - 26)) bits += cabq_resolve(T, st, q, qn);     /* no queue may pass 64 bins */
            b = bn; cw = cw_n; bw = bw_n;
        }
        bits += cabq_resolve(T, st, q, qn);
        if (live) D[ctx] = (uint8_t)st;
    }
    (void)sbits;
#endif
    PROF_ADD(26, t_3);
    out.vbits = bits; out.bits = sbits;
    return out;
}
__device__ __forceinline__ void prim_cab_residual(const FrameDev &F, MBLocal *L, CabWalk &C, int commit_)
{
    (void)F;
    const CabBits r = cab_residual_walk(L, commit_, -1);
    C.vbits += r.vbits; C.bits += r.bits;
}
__device__ __forceinline__ void prim_cab_residual_part(const FrameDev &F, MBLocal *L, CabWalk &C, int i8)
{
    (void)F;
    const CabBits r = cab_residual_walk(L, 0, i8);
    C.vbits += r.vbits; C.bits += r.bits;
}

/* ---------------------------------------------------------------- CAVLC size of the macroblock layer */
__device__ static const uint8_t vlc_coeff0_len_dev[5] = {1, 2, 4, 6, 2};
#include "pcamv_vlc_dev.h"
__device__ __forceinline__ int cavlc_level_size(int level, int *suffix_len)      /* common/vlc.c:874-914, cavlc.c:63-107 */
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
PCAMV_DEV int size_se_of(int v) { return size_ue_of((unsigned)(v <= 0 ? -v * 2 : v * 2 - 1)); }
/* x264_macroblock_write_cavlc as a bit counter (encoder/cavlc.c:290-600; rdo.c:41-47): one residual block per lane (0..15 luma,
 * 16..23 chroma AC, 24 / 25 chroma DC).  First every lane counts its coefficients (what the neighbours' nC reads, cavlc.c:133),
 * then sizes its block with nC from the counts.  Leaves the counts in L->nzc. */
/* part >= 0: x264_partition_size_cavlc of 8x8 `part` of a P_8x8 macroblock instead (encoder/cavlc.c:621-661): the sub-partition's MV
 * differences, its luma blocks when it is coded, its two chroma AC blocks */
__device__ __forceinline__ int prim_cavlc_mb(const FrameDev &F, MBLocal *L, int part_ = -1)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE(), part = rfl(part_);
    const int cbp_luma = rfl(L->cbp_luma), cbp_chroma = rfl(L->cbp_chroma);
    int bits = 0;
    /* header: every lane the same walk, counted on lane 0 */
    {
        int hb = 0, mvp[2];
#define GMVD(idx, w) (predict_mv(L, idx, w, mvp), size_se_of(L->cmv[scan8_of(idx)][0] - mvp[0]) + size_se_of(L->cmv[scan8_of(idx)][1] - mvp[1]))
        if (part >= 0) {
            switch (L->sub_part[part]) {
            case PCAMV_D_L0_8x8: hb += GMVD(4 * part, 2); break;
            case PCAMV_D_L0_8x4: hb += GMVD(4 * part, 2); hb += GMVD(4 * part + 2, 2); break;
            case PCAMV_D_L0_4x8: hb += GMVD(4 * part, 1); hb += GMVD(4 * part + 1, 1); break;
            default: for (int k = 0; k < 4; k++) hb += GMVD(4 * part + k, 1); break;
            }
        } else if (L->i_type == PCAMV_P_8x8) {
            hb += size_ue_of(3);
            for (int i = 0; i < 4; i++) { const int sp = L->sub_part[i]; hb += size_ue_of(sp == PCAMV_D_L0_8x8 ? 0 : sp == PCAMV_D_L0_8x4 ? 1 : sp == PCAMV_D_L0_4x8 ? 2 : 3); }
            for (int i = 0; i < 4; i++)
                switch (L->sub_part[i]) {
                case PCAMV_D_L0_8x8: hb += GMVD(4 * i, 2); break;
                case PCAMV_D_L0_8x4: hb += GMVD(4 * i, 2); hb += GMVD(4 * i + 2, 2); break;
                case PCAMV_D_L0_4x8: hb += GMVD(4 * i, 1); hb += GMVD(4 * i + 1, 1); break;
                default: for (int k = 0; k < 4; k++) hb += GMVD(4 * i + k, 1); break;
                }
        } else if (L->i_partition == PCAMV_D_16x16) { hb += size_ue_of(0); hb += GMVD(0, 4); }
        else if (L->i_partition == PCAMV_D_16x8) { hb += size_ue_of(1); hb += GMVD(0, 4); hb += GMVD(8, 4); }
        else { hb += size_ue_of(2); hb += GMVD(0, 2); hb += GMVD(4, 2); }
#undef GMVD
        if (part < 0) {
            hb += size_ue_of(vlc_inter_cbp_golomb_dev[(cbp_chroma << 4) | cbp_luma]);
            if (cbp_luma | cbp_chroma) hb += 1;          /* mb_qp_delta = 0 */
        }
        if (lane == 0) bits = hb;
    }
    /* which block this lane codes, and whether it is coded at all */
    const int idx = lane < 24 ? lane : 25 + (lane - 24);
    const bool mine = part >= 0 ? (lane < 16 ? (lane >> 2) == part && ((cbp_luma >> part) & 1) != 0 : lane < 24 && ((lane - 16) & 3) == part)
                    : lane < 26 && (lane < 16 ? ((cbp_luma >> (lane >> 2)) & 1) : lane < 24 ? (cbp_chroma & 2) != 0 : cbp_chroma != 0);
    const int count = lane < 16 ? 16 : lane < 24 ? 15 : 4;
    const int p8 = scan8_all_of(lane < 26 ? idx : 0);
    int level[16], total = 0, last = -1, trailing = 0;
    const bool flag = mine && L->nzc[p8] != 0;
    if (flag) {
        const int16_t *l = lane < 16 ? L->coef[lane] : lane < 24 ? L->coef[lane] + 1 : L->cdc[lane - 24];
#pragma unroll
        for (int i = 15; i >= 0; i--) {
            const int v = i < count ? l[i] : 0;
            if (v) { if (last < 0) last = i; level[total & 15] = v; total++; }
        }
    }
    PCAMV_WAVE_SYNC();
    if (flag) L->nzc[p8] = (uint8_t)total;
    PCAMV_WAVE_SYNC();
    if (mine) {
        int nC = 4;
        if (lane < 24) { int r = L->nzc[p8 - 1] + L->nzc[p8 - 8]; if (r < 0x80) r = (r + 1) >> 1; r &= 0x7f; nC = r < 2 ? 0 : r < 4 ? 1 : r < 8 ? 2 : 3; }
        if (!flag) bits += vlc_coeff0_len_dev[nC];
        else {
            const int16_t *l = lane < 16 ? L->coef[lane] : lane < 24 ? L->coef[lane] + 1 : L->cdc[lane - 24];
            while (trailing < 3 && trailing < total && iabs(level[trailing]) == 1) trailing++;
            bits += vlc_coeff_len_dev[nC * 64 + total * 4 + trailing - 4] + trailing;
            int sl = total > 10 && trailing < 3;
            if (trailing < total) {
                int v = level[trailing], s1 = sl, s2 = sl;
                if (trailing < 3) v -= v < 0 ? -1 : 1;
                bits += cavlc_level_size(v, &s1);
                cavlc_level_size(level[trailing], &s2); sl = s2;
                for (int i = trailing + 1; i < total; i++) bits += cavlc_level_size(level[i], &sl);
            }
            int total_zero = last + 1 - total;
            if (total < count) bits += lane >= 24 ? vlc_total_zeros_dc_len_dev[(total - 1) * 4 + total_zero] : vlc_total_zeros_len_dev[(total - 1) * 16 + total_zero];
            /* run_before: zeros between consecutive levels, from the top */
            int pos = last;
            for (int i = 0; i < total - 1 && total_zero > 0; i++) {
                int run = 0;
                while (--pos >= 0 && !l[pos]) run++;
                bits += vlc_run_before_len_dev[imin(total_zero - 1, 6) * 16 + run];
                total_zero -= run;
            }
        }
    }
    const int sum = wave_sum_all(bits);
    (void)F;
    PCAMV_WAVE_SYNC();
    return sum;
}

__device__ __forceinline__ int prim_cavlc_part8(const FrameDev &F, MBLocal *L, int i8) { return prim_cavlc_mb(F, L, i8); }

/* ---------------------------------------------------------------- x264_rd_cost_part for one 8x8 of a P_8x8 macroblock */
/* x264_macroblock_encode_p8x8 (encoder/macroblock.c:929-1052): prediction with the sub-partition's MVs (the cache), the 8x8's four
 * luma blocks with their own decimation rule (dropped below a score of 4), its two chroma 4x4 blocks without their DC; levels,
 * non-zero flags and reconstruction stay.  One block per lane like prim_mb_transform: the 8x8's luma blocks are the quad of
 * lanes 4 i8 .. 4 i8 + 3, its chroma blocks lanes 16 + i8 and 20 + i8. */
__device__ __forceinline__ void prim_encode_p8x8(const FrameDev &F, MBLocal *L, int i8_)
{
    const int i8 = rfl(i8_);
    prim_predict_mb(F, L, 0);               /* (the whole macroblock: only this 8x8's pixels are looked at) */
    const int lane = LANE();
    const bool is_l = lane < 16 && (lane >> 2) == i8, is_c = lane >= 16 && lane < 24 && ((lane - 16) & 3) == i8;
    const int ch = (lane - 16) >> 2, ci = (lane - 16) & 3;
    const int px = lane < 16 ? 4 * blk_x_of(lane) : ch * 8 + (ci & 1) * 4, py = lane < 16 ? 4 * blk_y_of(lane) : 16 + (ci >> 1) * 4;
    int16_t d[16];
    int nz = 0, score = 0, rawdc = 0;
    if (is_l || is_c) residual_block(F, L, px, py, lane < 16, d, &nz, &score, &rawdc, L->coef[lane < 24 ? lane : 0]);
    const int sc = (is_l && nz && F.b_dct_decimate) ? score : 0, nzl = is_l ? nz : 0;
    int q8 = sc + dpp_qp1(sc); q8 += dpp_qp2(q8);
    int any8 = nzl | dpp_qp1(nzl); any8 |= dpp_qp2(any8);
    const bool keep = any8 != 0 && !(F.b_dct_decimate && q8 < 4);
    uint8_t *dst = L->pred + py * 16 + px;
    if (is_l) { if (keep && nz) idct4x4_add(dst, d); L->nzc[scan8_of(lane)] = (uint8_t)(keep && nz); }
    else if (is_c) { if (nz) idct4x4_add(dst, d); L->nzc[scan8_all_of(lane)] = (uint8_t)nz; }
    const unsigned long long km = __ballot(is_l && keep);
    L->cbp_luma = (L->cbp_luma & ~(1 << i8)) | (km ? 1 << i8 : 0);
    L->cbp_chroma = 2;
    PCAMV_WAVE_SYNC();
}
/* ssd_plane( PIXEL_8x8, luma ) + ssd_plane( PIXEL_4x4, U / V ) of that 8x8 (rdo.c:106-128); psy: hadamard_ac of the 8x8 against the
 * source's energies of the same 8x8 (sum_satd / sum_sa8d, rdo.c:65-93), computed here rather than kept per block */
__device__ __forceinline__ int prim_ssd_part8(const FrameDev &F, MBLocal *L, int i8_)
{
    const int i8 = rfl(i8_);
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    const bool is_l = lane < 16 && (lane >> 2) == i8, is_c = lane >= 16 && lane < 24 && ((lane - 16) & 3) == i8;
    int v = 0;
    if (is_l || is_c) {
        const int ch = (lane - 16) >> 2, ci = (lane - 16) & 3;
        const int px = lane < 16 ? 4 * blk_x_of(lane) : ch * 8 + (ci & 1) * 4, py = lane < 16 ? 4 * blk_y_of(lane) : 16 + (ci >> 1) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t e = lds4(L->fenc + (py + k) * 16 + px), p = lds4(L->pred + (py + k) * 16 + px);
#pragma unroll
            for (int x = 0; x < 4; x++) { const int dd = (int)((e >> (8 * x)) & 255) - (int)((p >> (8 * x)) & 255); v += mul24s(dd, dd); }
        }
    }
    int ssd = wave_sum_all(v);
    if (F.psy_rd) {
        int s4, dc, s8, f4, fdc, f8;
        had_lane_sums(L->pred, lane, &s4, &dc, &s8);
        had_lane_sums(L->fenc, lane, &f4, &fdc, &f8);
        int fs = (f4 >> 1) - (fdc >> 1);
        if (!is_l) { s4 = 0; dc = 0; s8 = 0; fs = 0; fdc = 0; f8 = 0; }
        s4 = group_sum(s4, 16); dc = group_sum(dc, 16); s8 = group_sum(s8, 16); fs = group_sum(fs, 16); fdc = group_sum(fdc, 16); f8 = group_sum(f8, 16);
        const int S4 = __builtin_amdgcn_readlane(s4, 0), DC = __builtin_amdgcn_readlane(dc, 0), S8 = __builtin_amdgcn_readlane(s8, 0);
        const int FS = __builtin_amdgcn_readlane(fs, 0), FDC = __builtin_amdgcn_readlane(fdc, 0), F8 = __builtin_amdgcn_readlane(f8, 0);
        const int sum4 = (S4 - DC) >> 1, sum8 = (S8 - DC) >> 2, fsa8d = ((F8 + 2) >> 2) - (FDC >> 2);
        const int satd = (iabs(sum4 - FS) + iabs(sum8 - fsa8d)) >> 1;
        ssd += (satd * F.psy_rd * F.lambda + 128) >> 8;
    }
    PCAMV_WAVE_SYNC();
    return ssd;
}

/* what the coded macroblock leaves for its neighbours and successors: bottom row / right column of the non-zero cache, coded
 * block pattern, MV differences, the slice's context states -- stored write-through, like the motion the search hands over */
/* the trial just priced is the best so far: keep everything the final encode + entropy walk of that mode would produce again */
__device__ __forceinline__ void prim_rd_keep(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { const uint32_t *s = (const uint32_t *)L->pred; uint32_t *d = (uint32_t *)L->snap_pred; d[lane] = s[lane]; if (lane < 32) d[64 + lane] = s[64 + lane]; }
    if (lane < 12) ((uint32_t *)L->snap_nzc)[lane] = ((const uint32_t *)L->nzc)[lane];
    if (lane < 48) ((uint32_t *)L->snap_cmvd)[lane] = ((const uint32_t *)L->cmvd)[lane];
    if (F.b_cabac) {
        const uint32_t *s = (const uint32_t *)L_CABT(L); uint32_t *d = (uint32_t *)L_CABK(L);
        d[lane] = s[lane]; if (lane < PCAMV_CAB_USED / 4 - 64) d[64 + lane] = s[64 + lane];
    }
    if (lane == 0) { L->snap_cbp_luma = L->cbp_luma; L->snap_cbp_chroma = L->cbp_chroma; L->snap_nnz_mask = L->nnz_mask; }
    PCAMV_WAVE_SYNC();
}
/* the kept trial is the macroblock as decided: put its products where the final encode + walk would have left them (the context
 * states become the slice's; the mb_skip_flag decision, which a size trial does not contain, is the caller's) */
__device__ __forceinline__ void prim_rd_restore(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { const uint32_t *s = (const uint32_t *)L->snap_pred; uint32_t *d = (uint32_t *)L->pred; d[lane] = s[lane]; if (lane < 32) d[64 + lane] = s[64 + lane]; }
    if (lane < 12) ((uint32_t *)L->nzc)[lane] = ((const uint32_t *)L->snap_nzc)[lane];
    if (lane < 48) ((uint32_t *)L->cmvd)[lane] = ((const uint32_t *)L->snap_cmvd)[lane];
    if (F.b_cabac) {
        const uint32_t *s = (const uint32_t *)L_CABK(L); uint32_t *d = (uint32_t *)L_CAB(L, 0);
        d[lane] = s[lane]; if (lane < PCAMV_CAB_USED / 4 - 64) d[64 + lane] = s[64 + lane];
    }
    if (lane == 0) { L->cbp_luma = L->snap_cbp_luma; L->cbp_chroma = L->snap_cbp_chroma; L->nnz_mask = L->snap_nnz_mask; }
    PCAMV_WAVE_SYNC();
}
__device__ __forceinline__ void prim_rd_commit(const FrameDev &F, MBLocal *L, int skip_)
{
    const int skip = rfl(skip_);
    const int lane = LANE();
    const int xy = L->mb_xy;
    PCAMV_WAVE_SYNC();
    if (lane < 16) {
        const int k = lane & 7, is_right = lane >> 3;
        const int blk = is_right ? (k < 4 ? (k == 0 ? 5 : k == 1 ? 7 : k == 2 ? 13 : 15) : 17 + 2 * (k - 4)) : (k < 4 ? (k == 0 ? 10 : k == 1 ? 11 : k == 2 ? 14 : 15) : k < 6 ? 18 + (k - 4) : 22 + (k - 6));
        NB_ST8(&F.nb_nz[xy * 16 + lane], skip ? 0 : L->nzc[scan8_all_of(blk)]);
    } else if (lane < 24) {
        const int k = lane - 16, pos = k < 4 ? SCAN8_0 + k + 8 * 3 : SCAN8_0 + 3 + 8 * (k - 4);
        NB_ST32(&F.nb_mvd[(xy * 8 + k) * 2], skip ? 0u : ((const uint32_t *)L->cmvd)[pos]);
    } else if (lane == 24) {
        const int cbp = skip ? 0 : ((F.b_cabac ? (L->nzc[scan8_all_of(25)] << 9 | L->nzc[scan8_all_of(26)] << 10) : 0) | L->cbp_chroma << 4 | L->cbp_luma);
        NB_ST16(&F.nb_cbp[xy], cbp);
    }
    if (F.b_cabac) {
        const uint32_t *src = (const uint32_t *)L_CAB(L, 0);
        NB_ST32((uint32_t *)F.cabac + lane, src[lane]);
        if (lane < 52) NB_ST32((uint32_t *)F.cabac + 64 + lane, src[64 + lane]);
        if (F.dbg_hash && lane == 0) {
            uint32_t h = 2166136261u;
            for (int i = 0; i < 460; i++) h = (h ^ L_CAB(L, 0)[i]) * 16777619u;
            F.dbg_hash[xy] = h;
        }
    }
    if (F.inter & PCAMV_ANALYSE_PSUB8x8) {
        uint32_t *dst = (uint32_t *)F.cabac + PCAMV_CHAIN_NZ / 4;
        if (lane < 6) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) w |= (uint32_t)(skip ? 0 : L->nzc[scan8_all_of(4 * lane + k)]) << (8 * k);
            NB_ST32(dst + lane, w);
        } else if (lane < 22) NB_ST32(dst + lane, ((const uint32_t *)L->cmvd)[scan8_of(lane - 6)]);
    }
    PCAMV_WAVE_SYNC();
}
#endif
