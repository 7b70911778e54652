/*
 * pcamv_x264_glue.c -- the reference-side binding of libpcamv_gpu.so, as a maintainer would add it to the reference
 * (INTEGRATION.md walks through it).  Plain C against the reference's own headers (common/common.h: x264_t, h->info,
 * x264_frame_t) and include/pcamv_gpu.h; it keeps its state in a side structure so that it compiles against the
 * UNMODIFIED reference tree:
 *
 *     gcc -std=gnu99 -fsyntax-only -I/root/reference -I/root/reference/common -I/root/reference/encoder \
 *         -DHAVE_MALLOC_H -DARCH_X86_64 -DSYS_LINUX -Iinclude integration/pcamv_x264_glue.c
 *
 * (tests/test_integration_glue.py runs exactly that where /root/reference exists).  Where it hooks in:
 *   pcamv_glue_open     end of x264_encoder_open (encoder/encoder.c:630-832), after x264_validate_parameters
 *   pcamv_glue_pass1    x264_encoder_encode at do_encode: (encoder.c:2230-2390) for a P frame with embedding on, instead of
 *                       the first x264_slices_write: analysis + RCA costs + cover / cost assembly + STC on the GPU, results
 *                       into h->info exactly where pass 1 of the reference leaves them
 *   pcamv_glue_close    x264_encoder_close (encoder.c:2886)
 * Pass 2 stays the reference's own (types / partitions / MVs forced from h->info.cache, analyse.c:2870-3107).
 */
#include <string.h>
#include "common/common.h"
#include "encoder/ratecontrol.h"
#include "pcamv_gpu.h"

typedef struct {
    pcamv_ctx_t  *gpu;
    pcamv_mb_t   *mb;           /* mb_count records */
    pcamv_embed_t embed;        /* points at h->info.cover / rho_final / message / stego / filp */
} pcamv_glue_t;

int pcamv_glue_open( x264_t *h, pcamv_glue_t *g, int device )
{
    pcamv_params_t gp;
    memset( g, 0, sizeof(*g) );             /* a failed open leaves nothing for pcamv_glue_close to trip over */
    memset( &gp, 0, sizeof(gp) );
    gp.i_width  = 16 * h->sps->i_mb_width;
    gp.i_height = 16 * h->sps->i_mb_height;
    gp.i_me_method     = h->param.analyse.i_me_method;
    gp.i_me_range      = h->param.analyse.i_me_range;
    gp.i_subpel_refine = h->param.analyse.i_subpel_refine;
    gp.i_mv_range      = h->param.analyse.i_mv_range;           /* level-derived by then, encoder.c:558 */
    gp.b_chroma_me     = h->param.analyse.b_chroma_me;
    gp.b_fast_pskip    = h->param.analyse.b_fast_pskip;
    gp.b_dct_decimate  = h->param.analyse.b_dct_decimate;
    gp.b_cabac         = h->param.b_cabac;
    gp.inter           = h->param.analyse.inter & (X264_ANALYSE_I4x4 | X264_ANALYSE_PSUB16x16 | X264_ANALYSE_PSUB8x8);
    gp.i_chroma_qp_offset = h->pps->i_chroma_qp_index_offset;   /* psy-RD's -2 included, encoder.c:520-521 */
    gp.i_luma_deadzone[0] = h->param.analyse.i_luma_deadzone[0];
    gp.i_luma_deadzone[1] = h->param.analyse.i_luma_deadzone[1];
    gp.i_tscale = 256;                                          /* consecutive P frames, one reference */
    gp.i_psy_rd = h->mb.i_psy_rd;                               /* FIX8(f_psy_rd), 0 below subme 6 (encoder.c:513-515) */
    g->mb = x264_malloc( h->mb.i_mb_count * sizeof(pcamv_mb_t) );
    if( !g->mb || pcamv_gpu_open( &gp, device, &g->gpu ) < 0 )
    {
        x264_log( h, X264_LOG_ERROR, "pcamv_gpu_open failed (there is no CPU fallback)\n" );
        x264_free( g->mb );
        g->mb = NULL; g->gpu = NULL;
        return -1;
    }
    g->embed.cover = h->info.cover;  g->embed.rho = h->info.rho_final;  g->embed.message = h->info.message;
    g->embed.stego = h->info.stego;  g->embed.flip = h->info.filp;
    return 0;
}

void pcamv_glue_close( pcamv_glue_t *g )
{
    pcamv_gpu_close( g->gpu );            /* NULL after a failed open: a no-op */
    x264_free( g->mb );
    g->gpu = NULL; g->mb = NULL;
}

/* pass 1 of a P frame: h->fref0[0] is the (deblocked, expanded) reference, h->fenc the source picture */
int pcamv_glue_pass1( x264_t *h, pcamv_glue_t *g )
{
    x264_frame_t *ref = h->fref0[0];
    const uint8_t *rp[3] = { ref->plane[0], ref->plane[1], ref->plane[2] };
    const uint8_t *fp[3] = { h->fenc->plane[0], h->fenc->plane[1], h->fenc->plane[2] };
    /* the reference picture's final motion field feeds the temporal candidates (common/macroblock.c:444-467);
     * none when it is an I picture */
    const int16_t *pmv = ref->i_ref[0] > 0 ? &ref->mv[0][0][0] : NULL;
    const int8_t  *prf = ref->i_ref[0] > 0 ? ref->ref[0] : NULL;
    int i;
    if( pcamv_gpu_set_ref( g->gpu, rp, ref->i_stride, pmv, prf ) < 0 ||
        pcamv_gpu_upload_fenc( g->gpu, fp, h->fenc->i_stride ) < 0 ||
        pcamv_gpu_analyse_pframe( g->gpu, x264_ratecontrol_qp( h ), 1, g->mb, NULL ) < 0 ||
        pcamv_gpu_embed_pframe( g->gpu, h->param.eparam.iEmRate, NULL, 0, &g->embed ) < 0 )
    {
        x264_log( h, X264_LOG_ERROR, "pcamv: %s\n", pcamv_gpu_last_error( g->gpu ) );
        return -1;
    }
    /* the record -> h->info.cache[]: the same members under the same names (common/common.h:585-603); the orders differ
     * and pcamv_mb_t has no intra fields, so member by member.  cache[396] / the [6336] vectors must be sized
     * mb_count / 16 * mb_count for anything above CIF. */
    for( i = 0; i < h->mb.i_mb_count; i++ )
    {
        h->info.cache[i].i_type = g->mb[i].i_type;
        h->info.cache[i].i_partition = g->mb[i].i_partition;
        h->info.cache[i].i_qp = g->mb[i].i_qp;
        h->info.cache[i].used = g->mb[i].used;
        memcpy( h->info.cache[i].i_sub_partition, g->mb[i].i_sub_partition, sizeof(g->mb[i].i_sub_partition) );
        memcpy( h->info.cache[i].ref, g->mb[i].ref, sizeof(g->mb[i].ref) );
        memcpy( h->info.cache[i].mv, g->mb[i].mv, sizeof(g->mb[i].mv) );
        memcpy( h->info.cache[i].mv_stego, g->mb[i].mv_stego, sizeof(g->mb[i].mv_stego) );
        memcpy( h->info.cache[i].inter_stego_cost, g->mb[i].inter_stego_cost, sizeof(g->mb[i].inter_stego_cost) );
        memcpy( h->info.cache[i].pskip_mv_, g->mb[i].pskip_mv, sizeof(g->mb[i].pskip_mv) );
    }
    h->info.length = g->embed.n;
    h->info.num_filp = g->embed.num_flip;
    h->info.firstTime = 0;            /* x264_encoder_encode goes straight on to the second pass (encoder.c:2380-2390) */
    return 0;
}

/* pass 2 needs no search: x264_macroblock_analyse (analyse.c:2555) can skip everything above its "force from the record"
 * block (analyse.c:2870) when this holds -- the results of those searches are overwritten there anyway */
int pcamv_glue_pass2_skips_search( const x264_t *h )
{
    return h->info.embed_flag && !h->info.firstTime && h->sh.i_type == SLICE_TYPE_P;
}
