/*
 * pcamv_mbkernels.h -- per-macroblock bodies of the three analysis phases.
 *
 *   phase A  mbk_search : motion search + partition decision of one MB.  MB(x,y) needs the final
 *                         motion of its left, top-left, top and top-right neighbours
 *                         (common/macroblock.c:28-163, 422-439), so MBs on the anti-diagonal
 *                         x + 2y = d are independent and diagonals run in order.
 *   phase B  mbk_rca    : replacement-MV cost of ONE carrier MV (x264_ih_get_mv_cost,
 *                         analyse.c:2391-2550): 1 + up to 12 whole-MB re-encodes, 9 SATDs each.
 *                         Depends only on that MB's own decision -> embarrassingly parallel.
 *   phase C  mbk_encode : the pass-1 reconstruction of the MB (encoder/macroblock.c:484).
 */
#ifndef PCAMV_MBKERNELS_H
#define PCAMV_MBKERNELS_H
#include "pcamv_logic.h"

#ifdef PCAMV_HOST_EMU
#define PCAMV_LANE0 1
#define PCAMV_RFL(x) (x)
#else
#define PCAMV_LANE0 (LANE() == 0)
#define PCAMV_RFL(x) rfl(x)
#endif

/* what follows the decision: (--subme >= 6) the pass-1 reconstruction and the entropy coder's bookkeeping, then the record and the
 * macroblock's final motion, written through for the neighbours */
template <int TESA>
PCAMV_DEV void mbk_search_finish(const FrameDev &F, MBLocal *L, Analysis *a, int mb_x, int mb_y)
{
    if (MBRD_ON) {
        /* --subme >= 6: the macroblock as coded is part of what its neighbours read (reconstructed pixels for the intra
         * thresholds, non-zero flags / coded block pattern / MV differences / context states for the bit counts), so the
         * pass-1 reconstruction and the entropy coder's bookkeeping come before the hand-off */
        const unsigned long long t_c = PROF_T();
        L->b_skip_mc = 0;
        /* the trial of the decided mode has produced all of it already, unless no trial ran (P_SKIP, modes beyond the
         * thresholds) or the kept one is another mode */
        const int kept = L->i_type != PCAMV_P_SKIP && L->snap_part == L->i_partition;
        if (kept) prim_rd_restore(F, L);
        else mb_encode(F, L, 0, 1);
#ifdef PCAMV_HOST_EMU
        prim_store_rec(F, L);
#else
        prim_store_rec(F, L, true);
#endif
        if (PCAMV_LANE0 && F.nnz) F.nnz[L->mb_xy] = (uint16_t)L->nnz_mask;      /* with the pixels: what pass 2 takes over for a macroblock the embedding leaves alone */
        entropy_commit(F, L, kept);
        PROF_ADD(22, t_c);
    }
    const unsigned long long t_w = PROF_T();
    const int xy = L->mb_xy;
    int *slots = L->slots;
    const int used = F.embed && L->i_type != PCAMV_P_SKIP;
    const int n = carrier_slots(L->i_type, L->i_partition, L->sub_part, used, slots);
    pcamv_mb_t *r = &F.rec_mb[xy];
    const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w, b4 = 4 * (mb_y * s4 + mb_x), b8 = 2 * (mb_y * s8 + mb_x);
    PCAMV_WAVE_SYNC();
    /* the 16 per-4x4 entries of the record and of the frame's motion field (x264_macroblock_cache_save,
     * common/macroblock.c:1254-1364), one per lane */
    FOR_CAND(i, 16) {
        int i8 = scan8_of(i);
        r->ref[i] = L->cref[i8]; r->mv[i][0] = L->cmv[i8][0]; r->mv[i][1] = L->cmv[i8][1];
        r->mv_stego[i][0] = r->mv_stego[i][1] = 0; r->inter_stego_cost[i] = 0;
        int x = i & 3, y = i >> 2;
        NB_ST32(&F.mv[2 * (b4 + y * s4 + x)], NB_PACK16(L->cmv[SCAN8_0 + x + 8 * y][0], L->cmv[SCAN8_0 + x + 8 * y][1]));
    }
    if (PCAMV_LANE0) {
        r->i_type = L->i_type; r->i_partition = L->i_partition; r->i_qp = F.qp;
        for (int i = 0; i < 4; i++) r->i_sub_partition[i] = L->i_type == PCAMV_P_8x8 ? L->sub_part[i] : PCAMV_D_L0_8x8;
        r->pskip_mv[0] = L->pskip_mv[0]; r->pskip_mv[1] = L->pskip_mv[1];
        if (L->i_type != PCAMV_P_SKIP) { r->mvr16[0] = L->mvr_own[0]; r->mvr16[1] = L->mvr_own[1]; }
        else { r->mvr16[0] = r->mvr16[1] = 0; }
        r->used = (uint8_t)used; r->pad[0] = r->pad[1] = r->pad[2] = 0;
        for (int k = 0; k < n; k++) {
            MEState *me = slot_me(L, a, slots[k]);
            F.mvp_aux[(xy * 16 + slots[k]) * 2] = (int16_t)me->mvp[0];
            F.mvp_aux[(xy * 16 + slots[k]) * 2 + 1] = (int16_t)me->mvp[1];
        }
        NB_ST8(&F.mb_type[xy], L->i_type);
        NB_ST16(&F.ref8[b8], (uint16_t)(uint8_t)L->cref[scan8_of(0)] | (uint16_t)(uint8_t)L->cref[scan8_of(4)] << 8);
        NB_ST16(&F.ref8[b8 + s8], (uint16_t)(uint8_t)L->cref[scan8_of(8)] | (uint16_t)(uint8_t)L->cref[scan8_of(12)] << 8);
    }
    PROF_ADD(12, t_w);
}
template <int TESA>
PCAMV_DEV void mbk_search(const FrameDev &F, MBLocal *L, Analysis *a, int mb_x, int mb_y)
{
    const unsigned long long t_l = PROF_T();
    if ((TESA & 3) == 3 && F.b_mbrd) {
        /* --me tesa with the RD mode decision: the Hadamard exhaustive search keeps its survivor list in the LDS the RD stage keeps
         * the context states in (TESA_SLOT / L_CAB), so what the RD stage needs from memory is fetched after the searches, not with
         * the macroblock's other loads */
        mb_load(F, L, mb_x, mb_y, 0, 0);
        PROF_ADD(11, t_l);
        const int skip = analyse_s16<TESA>(F, L, a);
        if (!skip) analyse_s_rest<TESA>(F, L, a);
        { MbFetch pf; prim_mb_fetch(F, mb_x, mb_y, L->neighbour, 1, pf); prim_mb_fetch_store(F, L, 1, pf); }
        if (!skip) analyse_decide<TESA>(F, L, a);
        update_cache(L, a);
    } else {
        mb_load(F, L, mb_x, mb_y, 0, MBRD_ON);
        PROF_ADD(11, t_l);
        analyse_mb_search<TESA>(F, L, a);
    }
    mbk_search_finish<TESA>(F, L, a, mb_x, mb_y);
}

/* rebuild the decided partitioning (types, MVs, search-time mvp) from the record */
PCAMV_DEV int analysis_from_record(const FrameDev &F, MBLocal *L, Analysis *a, int xy, int *slots)
{
    const pcamv_mb_t *r = &F.rec_mb[xy];
    mb_load(F, L, xy % F.mb_w, xy / F.mb_w);
    L->i_type = r->i_type; L->i_partition = r->i_partition;
    for (int i = 0; i < 4; i++) L->sub_part[i] = r->i_sub_partition[i];
    const int n = carrier_slots(L->i_type, L->i_partition, L->sub_part, L->i_type != PCAMV_P_SKIP, slots);
    for (int k = 0; k < n; k++) {
        int s = slots[k], ip, xo, yo;
        MEState *me = slot_me(L, a, s);
        slot_geometry(L->i_type, L->i_partition, L->sub_part, s, &ip, &xo, &yo);
        me_setup(me, ip, xo, yo);
        me->mv[0] = r->mv[s][0]; me->mv[1] = r->mv[s][1];
        me->mvp[0] = F.mvp_aux[(xy * 16 + s) * 2]; me->mvp[1] = F.mvp_aux[(xy * 16 + s) * 2 + 1];
    }
    return n;
}

PCAMV_DEV void mbk_rca(const FrameDev &F, MBLocal *L, Analysis *a, int xy, int k)
{
    if (!F.rec_mb[xy].used) return;
    int *slots = L->slots;
    const int n = analysis_from_record(F, L, a, xy, slots);
    if (k >= n) return;
    MEState *me = slot_me(L, a, slots[k]);
    int dx = 0, dy = 0;
    const int bx = me->mv[0], by = me->mv[1];
    const int cost = rca_mv_cost(F, L, a, me, &dx, &dy, 0);
    if (PCAMV_LANE0) {
        pcamv_mb_t *r = &F.rec_mb[xy];
        r->mv_stego[slots[k]][0] = (int16_t)(bx + dx); r->mv_stego[slots[k]][1] = (int16_t)(by + dy);
        r->inter_stego_cost[slots[k]] = cost;
    }
}

/* phases C + B of one macroblock back to back (dataflow schedule): the pass-1 reconstruction, then
 * every carrier's replacement-MV cost.  The reconstruction of the macroblock as decided is also the
 * first re-encode of every carrier's RCA step, so it is made once (L->recb0); rca_mv_cost leaves the
 * decided MVs and the cache as it found them, so one rebuild of the analysis serves all carriers. */
/* fused = this wave has just searched this macroblock (dataflow schedule): MBLocal and Analysis still hold what
 * analysis_from_record would rebuild from the record (neighbour cache, limits, source pixels, decided MVs in the cache,
 * the carriers' search states), so only the fields that call resets are reset */
PCAMV_DEV int mbk_recon(const FrameDev &F, MBLocal *L, Analysis *a, int xy, int fused = 0, int have_rec = 0)
{
    const unsigned long long t_e = PROF_T();
    int n;
    if (fused) {
        L->b_skip_mc = 0;
        n = carrier_slots(L->i_type, L->i_partition, L->sub_part, L->i_type != PCAMV_P_SKIP, L->slots);
        for (int k = 0; k < n; k++) { MEState *me = slot_me(L, a, L->slots[k]); me->cost = me->cost_mv = me->cost_rec = 0; }
        if (have_rec) { PROF_ADD(10, t_e); return n; }      /* the search phase has reconstructed and stored the macroblock already (L->pred) */
    } else {
        n = analysis_from_record(F, L, a, xy, L->slots);
        update_cache(L, a);
    }
    mb_encode(F, L);
    prim_store_rec(F, L);
    if (PCAMV_LANE0 && F.nnz) F.nnz[xy] = (uint16_t)L->nnz_mask;
    PROF_ADD(10, t_e);
    return n;
}
PCAMV_DEV void mbk_rca_all(const FrameDev &F, MBLocal *L, Analysis *a, int xy, int n)
{
    int *slots = L->slots;
    if (F.rec_mb[xy].used && n > 0) {
        const unsigned long long t_w = PROF_T();
        PROF_CNT(42, n);
        prim_copy_pred(L, L->recb0);
        /* a 16x16 macroblock whose RCA neighbourhood (+-3 quarter pels) needs no MV clipping reads its
         * reference pixels from an LDS window loaded once, instead of ~30 scattered global fetches */
        int win = 0;
        if (L->i_type == PCAMV_P_L0 && L->i_partition == PCAMV_D_16x16) {
            const int bx = a->me16x16.mv[0], by = a->me16x16.mv[1];
            if (bx - 3 >= L->mv_min[0] && bx + 3 <= L->mv_max[0] && by - 3 >= L->mv_min[1] && by + 3 <= L->mv_max[1]) { prim_win_load(F, L, bx, by); win = 1; }
        }
        PROF_ADD(34, t_w);
        for (int k = 0; k < n; k++) {
            MEState *me = slot_me(L, a, slots[k]);
            int dx = 0, dy = 0;
            const int bx = me->mv[0], by = me->mv[1];
            const int cost = rca_mv_cost(F, L, a, me, &dx, &dy, 1, win);
            if (PCAMV_LANE0) {
                pcamv_mb_t *r = &F.rec_mb[xy];
                r->mv_stego[slots[k]][0] = (int16_t)(bx + dx); r->mv_stego[slots[k]][1] = (int16_t)(by + dy);
                r->inter_stego_cost[slots[k]] = cost;
            }
        }
    }
}
PCAMV_DEV void mbk_rca_encode(const FrameDev &F, MBLocal *L, Analysis *a, int xy, int fused = 0, int have_rec = 0)
{
#ifdef PCAMV_EXP_DBL_RCA      /* instruction-count experiment only */
    { const int n_ = mbk_recon(F, L, a, xy, fused, have_rec); mbk_rca_all(F, L, a, xy, n_); mbk_rca_all(F, L, a, xy, n_); return; }
#endif
    mbk_rca_all(F, L, a, xy, mbk_recon(F, L, a, xy, fused, have_rec));
}

/* pass 2 of one macroblock (analyse.c:2870-3107 + x264_macroblock_encode, semantics of DESIGN.md 5b): the
 * pass-1 type / partition, the record's MVs with mv_stego where the flip map says so, for a P_SKIP
 * macroblock the skip prediction from the FINAL neighbours; reconstruction; final motion + non-zero flags
 * for the loop filter and for the next frame's temporal candidates.  Same left / top / top-right
 * dependency as the search. */
/* store_rec = 0: the loop filter that follows in the same wave takes the reconstruction from L->pred and writes the
 * filtered macroblock itself */
/* unit: the second-pass kernel takes a run of macroblocks of a row per task and loads what they need of memory ONCE, together (k_pass2_deblock_flow,
 * P2Unit): the record, the index of the first carrier, "any carrier flipped", the first pass' non-zero flags come from there, and the pixels
 * of a macroblock the embedding left alone are already where the loop filter works -- nothing is loaded here then.  Returns 1 when the macroblock
 * was reconstructed anew (L->pred holds its pixels), 0 when the first pass' reconstruction stands. */
struct P2Pre { const pcamv_mb_t *r; int base, any_flip, nnz1, drain; };
PCAMV_DEV int mbk_pass2(const FrameDev &F, MBLocal *L, int mb_x, int mb_y, int store_rec = 1, const P2Pre *unit = nullptr)
{
    const int xy = mb_y * F.mb_w + mb_x;
    /* the macroblock's record (59 words) and the index of its first carrier come in with ONE memory round trip, into storage that is
     * idle in this pass (the candidate costs); read field by field from memory, type / partition / sub-partitions / used / MVs were a
     * dozen dependent round trips at the head of a task that is only ~3 k instructions long */
    const pcamv_mb_t *r = unit ? unit->r : (const pcamv_mb_t *)L->ccost;
    PCAMV_WAVE_SYNC();
#ifndef PCAMV_HOST_EMU
    /* asked for in the same round trip as the record, before anything is known about the macroblock: "is any carrier of it flipped" and
     * its first-pass reconstruction (used when the macroblock turns out to be what the first pass made, below: 7 of 8) */
    uint32_t pre_y = 0, pre_c = 0;
    int any_flip = 1;
    if (unit) any_flip = unit->any_flip;
    else {
        const int lane = LANE();
        if (F.rec_is_pass1) {
            pre_y = *(const uint32_t *)(F.rec[0] + (size_t)(mb_y * 16 + (lane >> 2)) * F.w + mb_x * 16 + (lane & 3) * 4);
            if (lane < 32) pre_c = *(const uint32_t *)(((lane >> 4) ? F.rec[2] : F.rec[1]) + (size_t)(mb_y * 8 + ((lane & 15) >> 1)) * (F.w >> 1) + mb_x * 8 + (lane & 1) * 4);
        }
        if (F.mbflip) any_flip = F.mbflip[xy];
    }
#else
    const int any_flip = 1;
#endif
    if (!unit) {
        FOR_CAND(i, (int)(sizeof(pcamv_mb_t) / 4) + 1) {
            if (i < (int)(sizeof(pcamv_mb_t) / 4)) ((uint32_t *)L->ccost)[i] = ((const uint32_t *)&F.rec_mb[xy])[i];
            else L->ccost[191] = F.car_base ? F.car_base[xy] : 0;
        }
    }
    PCAMV_WAVE_SYNC();
#ifndef PCAMV_HOST_EMU
    /* (a run of macroblocks per task: the skip prediction reads the left neighbour's final motion from memory, where this wave stored it a moment ago) */
    if (unit && unit->drain && r->i_type == PCAMV_P_SKIP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    /* only a skipped macroblock needs its neighbours (skip prediction); the source pixels only a macroblock that is re-encoded (below) */
    const int defer_fenc = unit != nullptr && r->i_type != PCAMV_P_SKIP;
    mb_load(F, L, mb_x, mb_y, r->i_type != PCAMV_P_SKIP ? (defer_fenc ? 2 : 1) : 0);
    L->i_type = r->i_type; L->i_partition = r->i_partition;
    for (int i = 0; i < 4; i++) L->sub_part[i] = r->i_sub_partition[i];
    cache_ref_set(L, 0, 0, 4, 4, 0);
    int same;
    if (L->i_type == PCAMV_P_SKIP) {
        L->i_partition = PCAMV_D_16x16;
        cache_mv_set(L, 0, 0, 4, 4, L->pskip_mv[0], L->pskip_mv[1]);
        same = L->pskip_mv[0] == r->pskip_mv[0] && L->pskip_mv[1] == r->pskip_mv[1];
    } else {
        int *slots = L->slots;
        const int n = carrier_slots(L->i_type, L->i_partition, L->sub_part, r->used, slots);
        const int base = unit ? unit->base : L->ccost[191];
        PCAMV_WAVE_SYNC();
        /* its carriers' flip flags: one more round trip, for the macroblocks that have a flipped carrier at all */
        if (PCAMV_RFL(any_flip)) { FOR_CAND(j, n) L->cxy[j] = F.flip ? (uint32_t)(F.flip[base + j] == 1) : 0u; }
        else { FOR_CAND(j, n) L->cxy[j] = 0u; }
        PCAMV_WAVE_SYNC();
        FOR_CAND(i, 16) {
            const int s = carrier_of_block(L->i_type, L->i_partition, L->sub_part, i);
            int flipped = 0;
            for (int j = 0; j < n; j++) if (slots[j] == s) flipped = (int)L->cxy[j];
            L->cmv[scan8_of(i)][0] = flipped ? r->mv_stego[s][0] : r->mv[i][0];
            L->cmv[scan8_of(i)][1] = flipped ? r->mv_stego[s][1] : r->mv[i][1];
        }
        PCAMV_WAVE_SYNC();
        same = 1;
        for (int j = 0; j < n; j++) if (L->cxy[j]) same = 0;
    }
    /* A macroblock whose motion is what the first pass decided -- no carrier of it flipped; skipped with the same skip prediction --
     * reconstructs to what the first pass stored (same type, motion, source, reference and quantiser): pixels and non-zero flags
     * are taken from there instead of being made again.  (~7 of 8 macroblocks at half a bit per carrier.) */
    const int reuse = same && F.rec_is_pass1;
    if (reuse) {
#ifdef PCAMV_HOST_EMU
        L->nnz_mask = F.nnz[xy];
#else
        L->nnz_mask = unit ? unit->nnz1 : rfl((int)F.nnz[xy]);
        if (!unit) {   /* (the layout of prim_store_rec) */
            const int lane = LANE();
            PCAMV_WAVE_SYNC();
            sts4(L->pred + (lane >> 2) * 16 + (lane & 3) * 4, pre_y);
            if (lane < 32) sts4(L->pred + 256 + ((lane & 15) >> 1) * 16 + (lane >> 4) * 8 + (lane & 1) * 4, pre_c);
            PCAMV_WAVE_SYNC();
        }
#endif
    } else {
        if (defer_fenc) prim_load_fenc(F, L);
        mb_encode(F, L);
#ifdef PCAMV_HOST_EMU
        prim_store_rec(F, L);
#else
        if (store_rec) prim_store_rec(F, L, true);
#endif
    }
    /* final motion, type and non-zero flags: read by the neighbours' skip prediction and loop filter in the same
     * launch, so stored write-through like the search's hand-off */
    const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w, b4 = 4 * (mb_y * s4 + mb_x), b8 = 2 * (mb_y * s8 + mb_x);
    PCAMV_WAVE_SYNC();
    FOR_CAND(i, 16) {
        int x = i & 3, y = i >> 2;
        NB_ST32(&F.mv[2 * (b4 + y * s4 + x)], NB_PACK16(L->cmv[SCAN8_0 + x + 8 * y][0], L->cmv[SCAN8_0 + x + 8 * y][1]));
    }
    if (PCAMV_LANE0) {
        NB_ST8(&F.mb_type[xy], L->i_type);
        NB_ST16(&F.ref8[b8], 0); NB_ST16(&F.ref8[b8 + s8], 0);
        NB_ST16(&F.nnz[xy], L->nnz_mask);
    }
    return !reuse;
}

PCAMV_DEV void mbk_encode(const FrameDev &F, MBLocal *L, Analysis *a, int xy)
{
    analysis_from_record(F, L, a, xy, L->slots);
    update_cache(L, a);
    mb_encode(F, L);
    prim_store_rec(F, L);
    if (PCAMV_LANE0 && F.nnz) F.nnz[xy] = (uint16_t)L->nnz_mask;
}
#endif
