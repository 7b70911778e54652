/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Drives the reference's OWN source files, compiled where they lie under
 * /root/reference, so that the CPU restatement in oracle/pcamv_oracle.c can be pinned
 * against what the reference actually computes.  Nothing from the reference is copied
 * here: this translation unit #includes encoder/analyse.c (to reach its static
 * functions: x264_macroblock_analyse, x264_ih_get_mv_cost, stc_embed via embed.h) and
 * links the reference's common/ *.c and encoder/{me,macroblock,cabac,cavlc,ratecontrol}.c
 * objects.  It does NOT link encoder/encoder.c or encoder/set.c: those need a generated
 * config.h and the absent third-party S-UNIWARD.lib (encoder.c:38,1441), which makes the
 * full encoder unbuildable here (DESIGN.md "Oracle").  What the harness therefore
 * replaces is only the *driver*: the few lines of x264_encoder_open (encoder.c:630-832)
 * and the raster MB loop of x264_slice_write (encoder.c:1240-1273,1938) that call the
 * reference functions in the reference's order, without entropy coding.
 *
 * Build: oracle/Makefile -> oracle/_ref/libpcamv_ref.so
 */
#include "encoder/analyse.c"   /* pulls common/common.h, me.h, rdo.c, embed.h, slicetype.c */

#include <stdlib.h>
#include <string.h>

/* encoder/rdo.c (pulled in by analyse.c) turns the bit writers into size counters with macros and leaves them
 * defined; this file calls the real writers of encoder/cabac.c / cavlc.c / common/bs.h again */
#undef bs_write1
#undef bs_write
#undef bs_write_ue
#undef bs_write_se
#undef bs_write_te
#undef x264_macroblock_write_cavlc
#undef x264_macroblock_write_cabac
#undef x264_cabac_encode_decision
#undef x264_cabac_encode_decision_noup
#undef x264_cabac_encode_terminal
#undef x264_cabac_encode_bypass
#undef x264_cabac_encode_ue_bypass
#undef x264_cabac_encode_flush

/* H.264 Table 8-15 (QPc as a function of qPI), padded by 12 on each side the way every
 * H.264 codec indexes it with a chroma offset in [-12,12]. Spec data, not reference text. */
static const uint8_t refh_chroma_qp_tab[52 + 24] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
    27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39,
    39, 39, 39, 39, 39, 39, 39, 39, 39, 39, 39, 39
};

typedef struct {
    x264_t *h;
    int width, height, mb_w, mb_h;
    int have_prev;
    uint32_t *dbg_state_hash;      /* optional [n_mb]: FNV-1a of the 460 CABAC context states after each macroblock was written */
    int dbg_mb; uint8_t *dbg_state;  /* optional: the 460 states after macroblock dbg_mb */
    int tail_skip;                   /* CAVLC: skipped macroblocks at the end of the frame coded last (their run is written with the slice's end) */
    int dbg_rd_mb, dbg_rd_n; const int32_t *dbg_rd_in; int32_t *dbg_rd_out;   /* optional: RD cost of given candidates at one macroblock */
} refh_t;

static void refh_log(void *p, int level, const char *fmt, va_list ap) { (void)p; if (level <= X264_LOG_WARNING) vfprintf(stderr, fmt, ap); }

/* mbcmp_init (encoder.c:615-625) is static inside encoder.c, which cannot be linked;
 * these are the same table selections expressed through the public vtable fields. */
static void refh_bind_cmp(x264_t *h)
{
    int satd = !h->mb.b_lossless && h->param.analyse.i_subpel_refine > 1;
    memcpy(h->pixf.mbcmp, satd ? h->pixf.satd : h->pixf.sad_aligned, sizeof(h->pixf.mbcmp));
    memcpy(h->pixf.mbcmp_unaligned, satd ? h->pixf.satd : h->pixf.sad, sizeof(h->pixf.mbcmp_unaligned));
    h->pixf.intra_mbcmp_x3_16x16 = satd ? h->pixf.intra_satd_x3_16x16 : h->pixf.intra_sad_x3_16x16;
    satd &= h->param.analyse.i_me_method == X264_ME_TESA;
    memcpy(h->pixf.fpelcmp, satd ? h->pixf.satd : h->pixf.sad, sizeof(h->pixf.fpelcmp));
    memcpy(h->pixf.fpelcmp_x3, satd ? h->pixf.satd_x3 : h->pixf.sad_x3, sizeof(h->pixf.fpelcmp_x3));
    memcpy(h->pixf.fpelcmp_x4, satd ? h->pixf.satd_x4 : h->pixf.sad_x4, sizeof(h->pixf.fpelcmp_x4));
}

/* Open a context. The arguments are the already-validated parameter values the reference
 * would hold after x264_validate_parameters (encoder.c:342-613) for a CQP, 1-ref,
 * no-B-frame, progressive encode: the caller passes mv_range (level-derived there). */
void *refh_open(int width, int height, int qp, int me_method, int me_range, int subme,
                int mv_range, int b_cabac, int embed, int inter_flags, int psy_rd_fix8, int chroma_qp_offset)
{
    if (width % 16 || height % 16) return NULL;
    int mb_w = width / 16, mb_h = height / 16;
    if (mb_w * mb_h > 396) return NULL;            /* h->info.cache[396], common.h:603 */
    refh_t *c = calloc(1, sizeof(*c));
    x264_t *h = x264_malloc(sizeof(x264_t));
    memset(h, 0, sizeof(x264_t));
    c->h = h; c->width = width; c->height = height; c->mb_w = mb_w; c->mb_h = mb_h;

    x264_param_default(&h->param);
    h->param.cpu = 0;
    h->param.pf_log = refh_log;
    h->param.i_width = width; h->param.i_height = height;
    h->param.rc.i_rc_method = X264_RC_CQP;
    h->param.rc.i_qp_constant = qp;
    h->param.rc.i_aq_mode = 0;                       /* encoder.c:436 */
    h->param.b_cabac = b_cabac;
    h->param.analyse.i_me_method = me_method;
    h->param.analyse.i_me_range = me_range;
    h->param.analyse.i_subpel_refine = subme;
    h->param.analyse.i_mv_range = mv_range;
    h->param.analyse.i_mv_range_thread = -1;
    h->param.analyse.b_transform_8x8 = 0;
    h->param.analyse.inter = inter_flags >= 0 ? (unsigned)inter_flags
                           : (X264_ANALYSE_I4x4 | X264_ANALYSE_PSUB16x16 | X264_ANALYSE_BSUB16x16); /* encoder.c:495-502 */
    h->param.analyse.intra = X264_ANALYSE_I4x4;
    /* encoder.c:511-522: psy-RD only from subme 6 on; the caller passes the chroma QP offset the reference would hold
     * after x264_validate_parameters (psy-RD lowers it by 1 or 2) */
    h->param.analyse.f_psy_rd = subme < 6 ? 0 : psy_rd_fix8 / 256.0f;
    h->mb.i_psy_rd = subme < 6 ? 0 : psy_rd_fix8; h->mb.i_psy_trellis = 0;
    h->param.analyse.i_chroma_qp_offset = chroma_qp_offset;
    h->param.i_bframe_adaptive = X264_B_ADAPT_NONE; /* encoder.c:464-465 */
    h->param.eparam.iEmRate = embed ? 0.5 : 0;

    h->sps = &h->sps_array[0];
    h->pps = &h->pps_array[0];
    h->sps->i_mb_width = mb_w; h->sps->i_mb_height = mb_h;
    h->sps->b_frame_mbs_only = 1;
    h->pps->i_chroma_qp_index_offset = chroma_qp_offset;      /* x264_pps_init, encoder/set.c */
    h->pps->i_cqm_preset = X264_CQM_FLAT;
    for (int i = 0; i < 6; i++) h->pps->scaling_list[i] = x264_cqm_flat16;
    if (x264_cqm_init(h) < 0) return NULL;

    h->mb.i_mb_count = mb_w * mb_h;
    h->frames.b_have_lowres = 0;
    h->frames.b_have_sub8x8_esa = !!(h->param.analyse.inter & X264_ANALYSE_PSUB8x8);
    h->frames.i_max_ref0 = 1;
    h->chroma_qp_table = refh_chroma_qp_tab + 12 + h->pps->i_chroma_qp_index_offset;

    x264_rdo_init();
    x264_predict_16x16_init(0, h->predict_16x16);
    x264_predict_8x8c_init(0, h->predict_8x8c);
    x264_predict_8x8_init(0, h->predict_8x8, &h->predict_8x8_filter);
    x264_predict_4x4_init(0, h->predict_4x4);
    x264_init_vlc_tables();
    x264_pixel_init(0, &h->pixf);
    x264_dct_init(0, &h->dctf);
    x264_zigzag_init(0, &h->zigzagf, 0);
    x264_mc_init(0, &h->mc);
    x264_quant_init(h, 0, &h->quantf);
    x264_deblock_init(0, &h->loopf);
    x264_dct_init_weights();
    refh_bind_cmp(h);

    h->thread[0] = h;
    h->fdec = x264_frame_new(h);
    h->fenc = x264_frame_new(h);
    h->fref0[0] = x264_frame_new(h);
    h->i_ref0 = 1;
    if (x264_macroblock_cache_init(h) < 0) return NULL;
    if (x264_ratecontrol_new(h) < 0) return NULL;
    h->out.i_bitstream = 1 << 20;
    h->out.p_bitstream = x264_malloc(h->out.i_bitstream);
    return c;
}

static void put_plane(uint8_t *dst, int dstride, const uint8_t *src, int w, int hgt)
{
    for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * dstride, src + (size_t)y * w, w);
}

/* Install the reference picture: raw reconstructed Y/U/V, then the reference's own border
 * expansion + 6-tap half-pel filter + integral image (encoder.c:1038-1047 whole-frame form). */
void refh_set_ref(void *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                  const int16_t *prev_mv /* [mb_h*4][mb_w*4][2] or NULL */,
                  const int8_t *prev_ref /* [mb_h*2][mb_w*2] or NULL */,
                  const int8_t *prev_type /* [mb] or NULL */)
{
    refh_t *c = ctx; x264_t *h = c->h; x264_frame_t *f = h->fref0[0];
    put_plane(f->plane[0], f->i_stride[0], y, c->width, c->height);
    put_plane(f->plane[1], f->i_stride[1], u, c->width / 2, c->height / 2);
    put_plane(f->plane[2], f->i_stride[2], v, c->width / 2, c->height / 2);
    h->sh.b_mbaff = 0;
    x264_frame_expand_border(h, f, 0, 1);
    x264_frame_filter(h, f, 0, 1);
    x264_frame_expand_border_filtered(h, f, 0, 1);
    f->i_poc = 0; f->b_kept_as_ref = 1;
    c->have_prev = prev_mv != NULL;
    if (prev_mv) {
        memcpy(f->mv[0], prev_mv, (size_t)h->mb.i_mb_count * 16 * 2 * sizeof(int16_t));
        memcpy(f->ref[0], prev_ref, (size_t)h->mb.i_mb_count * 4);
        if (prev_type) memcpy(f->mb_type, prev_type, h->mb.i_mb_count);
        else memset(f->mb_type, P_L0, h->mb.i_mb_count);
        f->i_ref[0] = 1; f->ref_poc[0][0] = -2;
        f->inv_ref_poc[0] = (256 + 1) / 2;              /* common/macroblock.c setup_inverse_delta_pocs, delta 2 */
    } else {
        f->i_ref[0] = 0;
        memset(f->mb_type, I_16x16, h->mb.i_mb_count);
        memset(f->ref[0], -1, (size_t)h->mb.i_mb_count * 4);
        memset(f->mv[0], 0, (size_t)h->mb.i_mb_count * 16 * 2 * sizeof(int16_t));
    }
}

void refh_set_fenc(void *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    refh_t *c = ctx; x264_frame_t *f = c->h->fenc;
    put_plane(f->plane[0], f->i_stride[0], y, c->width, c->height);
    put_plane(f->plane[1], f->i_stride[1], u, c->width / 2, c->height / 2);
    put_plane(f->plane[2], f->i_stride[2], v, c->width / 2, c->height / 2);
    f->i_frame = 1; f->i_poc = 2;
}

/* Copy out the reference picture's 4 luma planes (full,H,V,HV) + integral incl. padding. */
int refh_ref_stride(void *ctx) { return ((refh_t *)ctx)->h->fref0[0]->i_stride[0]; }
int refh_ref_cstride(void *ctx) { return ((refh_t *)ctx)->h->fref0[0]->i_stride[1]; }
void refh_get_ref_planes(void *ctx, uint8_t *out4 /* 4 * stride*(lines+64) */, uint16_t *integral)
{
    refh_t *c = ctx; x264_t *h = c->h; x264_frame_t *f = h->fref0[0];
    size_t sz = (size_t)f->i_stride[0] * (f->i_lines[0] + 2 * PADV);
    memcpy(out4, f->buffer[0], 4 * sz);
    if (integral && f->integral) memcpy(integral, f->buffer[3], sz * sizeof(uint16_t));
}

/* What x264_slice_write does around every macroblock besides analyse / encode (the harness replaces only that driver):
 * at the start of a macroblock row x264_fdec_filter_row (encoder.c:1019-1030) keeps the last, still unfiltered line of the
 * row above for intra prediction; after x264_macroblock_encode the macroblock goes through the entropy coder
 * (encoder.c:1900-1927), which adapts the CABAC context states (and, for CAVLC, leaves the coefficient counts in the
 * non-zero cache) that the RD mode decision of the following macroblocks reads (--subme >= 6, encoder/rdo.c:139-171). */
static void refh_row_start(x264_t *h, int mb_y)
{
    if (mb_y <= 0) return;
    for (int i = 0; i < 3; i++)
        memcpy(h->mb.intra_border_backup[0][i], h->fdec->plane[i] + ((mb_y * 16 >> !!i) - 1) * h->fdec->i_stride[i], h->sps->i_mb_width * 16 >> !!i);
}
static void refh_entropy_write(refh_t *c, int mb_xy, int *i_skip)
{
    x264_t *h = c->h;
    if (h->param.b_cabac) {
        if (mb_xy > h->sh.i_first_mb) x264_cabac_encode_terminal(&h->cabac);
        if (IS_SKIP(h->mb.i_type)) x264_cabac_mb_skip(h, 1);
        else { x264_cabac_mb_skip(h, 0); x264_macroblock_write_cabac(h, &h->cabac); }
        if (h->cabac.p > h->cabac.p_end - 4096) x264_cabac_encode_init(&h->cabac, h->out.p_bitstream, h->out.p_bitstream + h->out.i_bitstream);
        if (c->dbg_state_hash) {
            uint32_t hsh = 2166136261u;
            for (int i = 0; i < 460; i++) hsh = (hsh ^ h->cabac.state[i]) * 16777619u;
            c->dbg_state_hash[mb_xy] = hsh;
        }
        if (c->dbg_state && mb_xy == c->dbg_mb) memcpy(c->dbg_state, h->cabac.state, 460);
    } else {
        if (IS_SKIP(h->mb.i_type)) (*i_skip)++;
        else { bs_write_ue(&h->out.bs, *i_skip); *i_skip = 0; x264_macroblock_write_cavlc(h, &h->out.bs); }
        if (h->out.bs.p > h->out.bs.p_end - 4096) bs_init(&h->out.bs, h->out.p_bitstream, h->out.i_bitstream);
    }
}
/* diagnostics: x264_rd_cost_mb of n candidates {type, partition, mv[16][2] in block order} = 34 ints each, evaluated at
 * macroblock mb right after its cache_load (before its real analysis, which rebuilds everything the trial touches) */
void refh_set_debug_rd(void *ctx, int mb, int n, const int32_t *in, int32_t *out)
{ refh_t *c = ctx; c->dbg_rd_mb = mb; c->dbg_rd_n = n; c->dbg_rd_in = in; c->dbg_rd_out = out; }
static void refh_dbg_rd_run(refh_t *c, int qp)
{
    x264_t *h = c->h; x264_mb_analysis_t a;
    x264_mb_analyse_init(h, &a, qp);
    x264_mb_cache_fenc_satd(h);
    for (int k = 0; k < c->dbg_rd_n; k++) {
        const int32_t *in = c->dbg_rd_in + 34 * k;
        if (in[0] < 0) {        /* intra SATD analysis with i_satd_inter = in[1] */
            x264_mb_analysis_t b; x264_mb_analyse_init(h, &b, qp);
            x264_mb_analyse_intra_chroma(h, &b);
            x264_mb_analyse_intra(h, &b, in[1] - b.i_satd_i8x8chroma);
            c->dbg_rd_out[4 * k] = b.i_satd_i16x16; c->dbg_rd_out[4 * k + 1] = b.i_satd_i4x4; c->dbg_rd_out[4 * k + 2] = b.i_satd_i8x8chroma; c->dbg_rd_out[4 * k + 3] = b.b_fast_intra;
            continue;
        }
        h->mb.i_type = in[0]; h->mb.i_partition = in[1];
        for (int i = 0; i < 4; i++) h->mb.i_sub_partition[i] = D_L0_8x8;
        for (int i = 0; i < 16; i++) {
            h->mb.cache.ref[0][x264_scan8[i]] = 0;
            h->mb.cache.mv[0][x264_scan8[i]][0] = in[2 + 2 * i]; h->mb.cache.mv[0][x264_scan8[i]][1] = in[3 + 2 * i];
        }
        c->dbg_rd_out[4 * k] = x264_rd_cost_mb(h, a.i_lambda2);
        c->dbg_rd_out[4 * k + 1] = ssd_mb(h);
        c->dbg_rd_out[4 * k + 2] = h->mb.i_cbp_luma | h->mb.i_cbp_chroma << 4;
        c->dbg_rd_out[4 * k + 3] = 0;
    }
}
void refh_set_debug(void *ctx, uint32_t *state_hash, int dump_mb, uint8_t *dump_state)
{ refh_t *c = ctx; c->dbg_state_hash = state_hash; c->dbg_mb = dump_mb; c->dbg_state = dump_state; }

typedef struct {
    int32_t type, partition, qp;
    uint8_t sub_partition[4];
    int8_t  ref[16];
    int16_t mv[16][2];          /* x264 block-index order (analyse.c:2893-2898) */
    int16_t mv_stego[16][2];
    int32_t stego_cost[16];
    int16_t pskip_mv[2];
    int16_t mvr16[2];           /* 16x16 search result kept for neighbour candidates */
    uint8_t used, pad[3];
} refh_mb_t;

/* Pass-1 analysis of one P frame: the raster loop of x264_slice_write without bitstream. */
int refh_analyse_pframe(void *ctx, int qp, refh_mb_t *out,
                        uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v)
{
    refh_t *c = ctx; x264_t *h = c->h;
    int embed = h->param.eparam.iEmRate != 0;
    h->sh.i_type = SLICE_TYPE_P;
    h->sh.i_first_mb = 0; h->sh.i_last_mb = h->mb.i_mb_count;
    h->sh.b_mbaff = 0; h->sh.i_num_ref_idx_l0_active = 1; h->sh.i_qp = qp;
    h->i_ref0 = 1; h->i_ref1 = 0;
    h->mb.pic.i_fref[0] = 1; h->mb.pic.i_fref[1] = 0;
    h->fdec->i_poc = 2; h->fdec->b_kept_as_ref = 1; h->fdec->i_frame = 1;
    h->fenc->i_poc = 2;
    h->info.embed_flag = embed; h->info.firstTime = 1;
    x264_macroblock_slice_init(h);
    x264_ratecontrol_start(h, qp + 1);
    if (embed)
        for (int i = 0; i < 396; i++) {               /* same reset as encoder.c:1191-1198 */
            h->info.cache[i].used = 0; h->info.cache[i].i_type = 6;
            memset(h->info.cache[i].mv, 0, sizeof(int16_t) * 32);
            memset(h->info.cache[i].mv_stego, 0, sizeof(int16_t) * 32);
            memset(h->info.cache[i].inter_stego_cost, 0, sizeof(int) * 16);
            memset(h->info.cache[i].ref, -2, 16);
        }
    memset(&h->stat.frame, 0, sizeof(h->stat.frame));
    x264_cabac_context_init(&h->cabac, h->sh.i_type, h->sh.i_qp, 0);
    x264_cabac_encode_init(&h->cabac, h->out.p_bitstream, h->out.p_bitstream + h->out.i_bitstream);
    h->mb.i_last_qp = qp; h->mb.i_last_dqp = 0;

    bs_init(&h->out.bs, h->out.p_bitstream, h->out.i_bitstream);
    int i_skip = 0;
    for (int my = 0; my < c->mb_h; my++)
        for (int mx = 0; mx < c->mb_w; mx++) {
            int mb_xy = my * c->mb_w + mx;
            if (mx == 0) refh_row_start(h, my);
            x264_macroblock_cache_load(h, mx, my);
            if (c->dbg_rd_out && mb_xy == c->dbg_rd_mb) refh_dbg_rd_run(c, qp);
            refh_mb_t *o = &out[mb_xy];
            memset(o, 0, sizeof(*o));
            o->pskip_mv[0] = h->mb.cache.pskip_mv[0]; o->pskip_mv[1] = h->mb.cache.pskip_mv[1];
            x264_macroblock_analyse(h);
            x264_macroblock_encode(h);
            refh_entropy_write(c, mb_xy, &i_skip);
            o->type = h->mb.i_type; o->partition = h->mb.i_partition; o->qp = h->mb.i_qp;
            /* i_sub_partition is stale outside P_8x8 and mvr is never written for an early P_SKIP
             * (analyse.c:1170-1177 returns first): report neutral values for those don't-cares */
            if (h->mb.i_type == P_8x8) memcpy(o->sub_partition, h->mb.i_sub_partition, 4);
            else memset(o->sub_partition, D_L0_8x8, 4);
            for (int i = 0; i < 16; i++) {
                o->ref[i] = h->mb.cache.ref[0][x264_scan8[i]];
                o->mv[i][0] = h->mb.cache.mv[0][x264_scan8[i]][0];
                o->mv[i][1] = h->mb.cache.mv[0][x264_scan8[i]][1];
            }
            if (h->mb.i_type != P_SKIP) { o->mvr16[0] = h->mb.mvr[0][0][mb_xy][0]; o->mvr16[1] = h->mb.mvr[0][0][mb_xy][1]; }
            if (embed) {
                o->used = h->info.cache[mb_xy].used;
                memcpy(o->mv_stego, h->info.cache[mb_xy].mv_stego, sizeof(o->mv_stego));
                memcpy(o->stego_cost, h->info.cache[mb_xy].inter_stego_cost, sizeof(o->stego_cost));
            }
            x264_macroblock_cache_save(h);
        }
    c->tail_skip = i_skip;
    x264_frame_t *f = h->fdec;
    for (int y = 0; y < c->height; y++) memcpy(rec_y + (size_t)y * c->width, f->plane[0] + (size_t)y * f->i_stride[0], c->width);
    for (int y = 0; y < c->height / 2; y++) {
        memcpy(rec_u + (size_t)y * c->width / 2, f->plane[1] + (size_t)y * f->i_stride[1], c->width / 2);
        memcpy(rec_v + (size_t)y * c->width / 2, f->plane[2] + (size_t)y * f->i_stride[2], c->width / 2);
    }
    return 0;
}

/* The slice data the reference's entropy coder has produced for the frame analysed / re-encoded last: with CABAC the bytes from
 * the first macroblock's mb_skip_flag to the end_of_slice terminal and the flush (x264_cabac_encode_flush, as x264_slice_write
 * ends a slice, encoder.c:1332), i.e. what follows the slice header and its alignment bits.  Golden input of the product's
 * MV-syntax extractor (tests/golden/pslice_*.npz).  Call once, after the frame. */
int refh_slice_data(void *ctx, uint8_t *buf, int cap)
{
    refh_t *c = ctx; x264_t *h = c->h;
    if (!h->param.b_cabac) {        /* CAVLC: the run of skipped macroblocks that ends the slice, then the rbsp trailing bits (encoder.c:1320-1340) */
        if (c->tail_skip > 0) bs_write_ue(&h->out.bs, c->tail_skip);
        c->tail_skip = 0;
        bs_rbsp_trailing(&h->out.bs);
        int nb = bs_pos(&h->out.bs) / 8;
        if (nb > cap) return -2;
        memcpy(buf, h->out.p_bitstream, nb);
        return nb;
    }
    x264_cabac_encode_flush(h, &h->cabac);
    int n = (int)(h->cabac.p - h->cabac.p_start);
    if (n > cap) return -2;
    memcpy(buf, h->cabac.p_start, n);
    return n;
}

/* Pass 2 of the fork's two-pass P frame (encoder.c:2380-2390 re-enters x264_slice_write with firstTime = 0):
 * x264_macroblock_analyse forces type / partition / MVs from h->info.cache and swaps in mv_stego where
 * h->info.filp says so (analyse.c:2574-2577, 2658-2679, 2870-3107), x264_macroblock_encode reconstructs,
 * then the loop filter runs over the frame (x264_fdec_filter_row -> x264_frame_deblock_row).  Must follow
 * refh_analyse_pframe on the same frame.  Dumps the final motion (incl. the skip MVs pass 2 predicts from
 * the final neighbours), per-4x4 non-zero flags, the reconstruction before and after deblocking. */
int refh_pass2_pframe(void *ctx, int qp, const int8_t *flips, int n_flips, refh_mb_t *out, uint8_t *nnz_out,
                      uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint8_t *dbk_y, uint8_t *dbk_u, uint8_t *dbk_v)
{
    refh_t *c = ctx; x264_t *h = c->h;
    if (n_flips < 0 || n_flips > 6336) return -1;
    h->info.firstTime = 0;
    h->info.i_mv_no = 0; h->info.num_mv_modify_real = 0;
    memset(h->info.filp, 0, sizeof(h->info.filp));
    memcpy(h->info.filp, flips, n_flips);
    h->sh.i_type = SLICE_TYPE_P;
    h->sh.i_disable_deblocking_filter_idc = 0; h->sh.i_alpha_c0_offset = 0; h->sh.i_beta_offset = 0;
    x264_macroblock_slice_init(h);
    x264_cabac_context_init(&h->cabac, h->sh.i_type, h->sh.i_qp, 0);
    x264_cabac_encode_init(&h->cabac, h->out.p_bitstream, h->out.p_bitstream + h->out.i_bitstream);
    h->mb.i_last_qp = qp; h->mb.i_last_dqp = 0;
    bs_init(&h->out.bs, h->out.p_bitstream, h->out.i_bitstream);
    int i_skip = 0;
    for (int my = 0; my < c->mb_h; my++)
        for (int mx = 0; mx < c->mb_w; mx++) {
            int mb_xy = my * c->mb_w + mx;
            if (mx == 0) refh_row_start(h, my);
            x264_macroblock_cache_load(h, mx, my);
            x264_macroblock_analyse(h);
            x264_macroblock_encode(h);
            refh_entropy_write(c, mb_xy, &i_skip);
            refh_mb_t *o = &out[mb_xy];
            memset(o, 0, sizeof(*o));
            o->type = h->mb.i_type; o->partition = h->mb.i_partition; o->qp = h->mb.i_qp;
            if (h->mb.i_type == P_8x8) memcpy(o->sub_partition, h->mb.i_sub_partition, 4);
            else memset(o->sub_partition, D_L0_8x8, 4);
            for (int i = 0; i < 16; i++) {
                o->ref[i] = h->mb.cache.ref[0][x264_scan8[i]];
                o->mv[i][0] = h->mb.cache.mv[0][x264_scan8[i]][0];
                o->mv[i][1] = h->mb.cache.mv[0][x264_scan8[i]][1];
                nnz_out[mb_xy * 16 + i] = h->mb.cache.non_zero_count[x264_scan8[i]];
            }
            o->pskip_mv[0] = h->mb.cache.pskip_mv[0]; o->pskip_mv[1] = h->mb.cache.pskip_mv[1];
            x264_macroblock_cache_save(h);
        }
    c->tail_skip = i_skip;
    x264_frame_t *f = h->fdec;
    for (int pass = 0; pass < 2; pass++) {
        uint8_t *py = pass ? dbk_y : rec_y, *pu = pass ? dbk_u : rec_u, *pv = pass ? dbk_v : rec_v;
        if (pass) for (int my = 0; my < c->mb_h; my++) x264_frame_deblock_row(h, my);
        for (int y = 0; y < c->height; y++) memcpy(py + (size_t)y * c->width, f->plane[0] + (size_t)y * f->i_stride[0], c->width);
        for (int y = 0; y < c->height / 2; y++) {
            memcpy(pu + (size_t)y * c->width / 2, f->plane[1] + (size_t)y * f->i_stride[1], c->width / 2);
            memcpy(pv + (size_t)y * c->width / 2, f->plane[2] + (size_t)y * f->i_stride[2], c->width / 2);
        }
    }
    return h->info.i_mv_no;
}

/* ---- primitive-level entry points (checkasm-style differential testing, tools/checkasm.c) ---- */
int refh_sad(void *ctx, int i_pixel, uint8_t *a, int sa, uint8_t *b, int sb) { return ((refh_t *)ctx)->h->pixf.sad[i_pixel](a, sa, b, sb); }
int refh_satd(void *ctx, int i_pixel, uint8_t *a, int sa, uint8_t *b, int sb) { return ((refh_t *)ctx)->h->pixf.satd[i_pixel](a, sa, b, sb); }
int refh_ssd(void *ctx, int i_pixel, uint8_t *a, int sa, uint8_t *b, int sb) { return ((refh_t *)ctx)->h->pixf.ssd[i_pixel](a, sa, b, sb); }
/* intra prediction into a stride-32 buffer that holds the neighbours (common/predict.c): kind 0 = 16x16, 1 = 8x8 chroma, 2 = 4x4 */
void refh_predict(void *ctx, int kind, int mode, uint8_t *dst)
{
    x264_t *h = ((refh_t *)ctx)->h;
    if (kind == 0) h->predict_16x16[mode](dst); else if (kind == 1) h->predict_8x8c[mode](dst); else h->predict_4x4[mode](dst);
}
void refh_hadamard_ac(void *ctx, int i_pixel, uint8_t *pix, int stride, uint32_t out[2])
{ uint64_t v = ((refh_t *)ctx)->h->pixf.hadamard_ac[i_pixel](pix, stride); out[0] = (uint32_t)v; out[1] = (uint32_t)(v >> 32); }
int refh_sa8d(void *ctx, int i_pixel, uint8_t *a, int sa, uint8_t *b, int sb) { return ((refh_t *)ctx)->h->pixf.sa8d[i_pixel](a, sa, b, sb); }
void refh_mc_luma(void *ctx, uint8_t *dst, int ds, uint8_t *src4[4], int ss, int mvx, int mvy, int w, int hgt)
{ ((refh_t *)ctx)->h->mc.mc_luma(dst, ds, src4, ss, mvx, mvy, w, hgt); }
void refh_mc_chroma(void *ctx, uint8_t *dst, int ds, uint8_t *src, int ss, int mvx, int mvy, int w, int hgt)
{ ((refh_t *)ctx)->h->mc.mc_chroma(dst, ds, src, ss, mvx, mvy, w, hgt); }
int refh_ads(void *ctx, int i_pixel, int enc_dc[4], uint16_t *sums, int delta, uint16_t *cost_mvx, int16_t *mvs, int width, int thresh)
{ return ((refh_t *)ctx)->h->pixf.ads[i_pixel](enc_dc, sums, delta, cost_mvx, mvs, width, thresh); }

/* encode one 16x16 residual the way x264_macroblock_encode's inter branch does, isolated:
 * fenc (stride 16), pred (stride 32) -> recon in pred. Returns cbp_luma. */
int refh_cost_mv_table(int qp, int16_t *out /* [4*4*2048+1] */)
{
    x264_mb_analysis_t a; x264_t hh; memset(&hh, 0, sizeof(hh)); memset(&a, 0, sizeof(a));
    a.i_qp = qp; a.i_lambda = x264_lambda_tab[qp];
    hh.sh.i_num_ref_idx_l0_active = 1; hh.param.analyse.i_me_method = X264_ME_HEX;
    x264_mb_analyse_load_costs(&hh, &a);
    memcpy(out, a.p_cost_mv - 2 * 4 * 2048, (4 * 4 * 2048 + 1) * sizeof(int16_t));
    return a.p_cost_ref0[0];
}

/* a single motion search through the reference's x264_me_search_ref (me.c:158) */
void refh_me_search(void *ctx, int qp, int mb_x, int mb_y, int i_pixel, int xoff, int yoff,
                    const int16_t mvp[2], const int16_t (*mvc)[2], int i_mvc, int16_t out_mv[2], int out_cost[2])
{
    refh_t *c = ctx; x264_t *h = c->h; x264_mb_analysis_t a; x264_me_t m;
    memset(&a, 0, sizeof(a)); memset(&m, 0, sizeof(m));
    h->sh.i_type = SLICE_TYPE_P; h->sh.i_first_mb = 0; h->sh.i_last_mb = h->mb.i_mb_count;
    h->sh.i_num_ref_idx_l0_active = 1; h->mb.pic.i_fref[0] = 1; h->i_ref0 = 1;
    x264_macroblock_slice_init(h);
    x264_macroblock_cache_load(h, 0, mb_y);     /* x==0 initialises the vertical limits (analyse.c:285) */
    x264_mb_analyse_init(h, &a, qp);
    x264_macroblock_cache_load(h, mb_x, mb_y);
    x264_mb_analyse_init(h, &a, qp);
    x264_mb_analyse_load_costs(h, &a);
    m.i_pixel = i_pixel; m.p_cost_mv = a.p_cost_mv; m.i_ref = 0; m.i_ref_cost = 0;
    m.i_stride[0] = h->mb.pic.i_stride[0]; m.i_stride[1] = h->mb.pic.i_stride[1];
    LOAD_FENC(&m, h->mb.pic.p_fenc, xoff, yoff);
    LOAD_HPELS(&m, h->mb.pic.p_fref[0][0], 0, 0, xoff, yoff);
    m.mvp[0] = mvp[0]; m.mvp[1] = mvp[1];
    int16_t lmvc[16][2];
    for (int i = 0; i < i_mvc; i++) { lmvc[i][0] = mvc[i][0]; lmvc[i][1] = mvc[i][1]; }
    x264_me_search_ref(h, &m, lmvc, i_mvc, NULL);
    out_mv[0] = m.mv[0]; out_mv[1] = m.mv[1]; out_cost[0] = m.cost; out_cost[1] = m.cost_mv;
}

/* syndrome-trellis embedding straight from the reference header embed.h:309 */
int refh_stc_embed(const uint8_t *cover, int n, const uint8_t *msg, int m, const float *rho, int hgt, uint8_t *stego)
{
    return stc_embed((uint8_t *)cover, n, (uint8_t *)msg, m, (float *)rho, stego, hgt);
}
