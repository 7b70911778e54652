/*
 * pcamv_gpu.h -- C ABI of the MI355X P-frame analysis + MV-steganography library
 *                (libpcamv_gpu.so, built from video-steganography-pcamv_amd/csrc/).
 *
 * Frame-granular replacement for the per-macroblock work the reference does inside
 * x264_slice_write() (encoder/encoder.c:1176-2011) on the first of its two passes over a
 * P frame:
 *
 *   entry point                 replaces (reference file:line)
 *   --------------------------  -----------------------------------------------------------
 *   pcamv_gpu_open              vtable + table set-up in x264_encoder_open, encoder.c:694-766
 *                               (x264_cqm_init common/set.c:68, cost tables analyse.c:193-229)
 *   pcamv_gpu_set_ref           x264_fdec_filter_row's plane production, encoder.c:1038-1047:
 *                               x264_frame_expand_border (common/frame.c:246),
 *                               x264_frame_filter/hpel_filter (common/mc.c:453,167),
 *                               x264_frame_expand_border_filtered (common/frame.c:275)
 *   pcamv_gpu_upload_fenc       x264_frame_copy_picture (common/frame.c:183-220)
 *   pcamv_gpu_analyse_pframe    pass 1 of the raster MB loop, encoder.c:1240-1273:
 *                               x264_macroblock_cache_load (common/macroblock.c:914),
 *                               x264_macroblock_analyse (encoder/analyse.c:2555-3697) incl.
 *                               x264_me_search_ref (encoder/me.c:158), refine_subpel (me.c:715),
 *                               x264_ih_get_mv_cost (analyse.c:2391) and the pass-1 record
 *                               (analyse.c:3518-3689); the pass-1 x264_macroblock_encode
 *                               (encoder/macroblock.c:484) reconstruction
 *   pcamv_gpu_embed_pframe      cover/cost assembly + message + stc_embed + flip map,
 *                               encoder.c:1561-1855, embed.h:309-548
 *   pcamv_gpu_stc_extract       (no reference counterpart: extractor defined in SURVEY 8(c))
 *   pcamv_gpu_close             x264_encoder_close's frees
 *
 * All functions return 0 on success and a negative PCAMV_E* code on error; the message is
 * available from pcamv_gpu_last_error().  One context per encoder, not re-entrant (the
 * reference's embedding path is single-threaded).  The caller owns every host
 * buffer it passes; the context owns all device memory.  There is NO CPU fallback: without a
 * HIP device every entry point fails with PCAMV_ENODEV.
 */
#ifndef PCAMV_GPU_H
#define PCAMV_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCAMV_EINVAL  (-1)
#define PCAMV_ENODEV  (-2)
#define PCAMV_ENOMEM  (-3)
#define PCAMV_EHIP    (-4)
#define PCAMV_EUNSUP  (-5)

/* enum values are the reference's own (x264.h:106-111, common/macroblock.h mb_class_e / mb_partition_e) */
enum { PCAMV_ME_DIA = 0, PCAMV_ME_HEX = 1, PCAMV_ME_UMH = 2, PCAMV_ME_ESA = 3, PCAMV_ME_TESA = 4 };
enum { PCAMV_P_L0 = 4, PCAMV_P_8x8 = 5, PCAMV_P_SKIP = 6 };
enum { PCAMV_D_L0_4x4 = 0, PCAMV_D_L0_8x4 = 1, PCAMV_D_L0_4x8 = 2, PCAMV_D_L0_8x8 = 3,
       PCAMV_D_8x8 = 13, PCAMV_D_16x8 = 14, PCAMV_D_8x16 = 15, PCAMV_D_16x16 = 16 };
#define PCAMV_ANALYSE_I4x4      0x0001u   /* x264.h:97: intra 4x4 SATD analysis (only its cost is used, from subme 6 on) */
#define PCAMV_ANALYSE_PSUB16x16 0x0010u   /* x264.h:99  */
#define PCAMV_ANALYSE_PSUB8x8   0x0020u   /* x264.h:100 */

/* The subset of x264_param_t (x264.h:154-311) this path reads, with the reference's field
 * names and its post-x264_validate_parameters meaning (encoder.c:342-613). */
typedef struct pcamv_params_t {
    int32_t i_width, i_height;        /* luma size, multiples of 16                              */
    int32_t i_me_method;              /* analyse.i_me_method  (PCAMV_ME_*, all five)             */
    int32_t i_me_range;               /* analyse.i_me_range   (default 16, common.c:121); <= 16 with PCAMV_ME_TESA */
    int32_t i_subpel_refine;          /* analyse.i_subpel_refine, 1..7 (6 and 7 are the same for P frames: RD mode decision, incl.
                                         x264_rd_cost_part for sub-8x8 partitions); 8 / 9 (RD refinement of MVs: off in the fork's P frames) PCAMV_EUNSUP */
    int32_t i_mv_range;               /* analyse.i_mv_range after level lookup, encoder.c:558    */
    int32_t b_chroma_me;              /* analyse.b_chroma_me  (default 1)                        */
    int32_t b_fast_pskip;             /* analyse.b_fast_pskip (default 1)                        */
    int32_t b_dct_decimate;           /* analyse.b_dct_decimate (default 1)                      */
    int32_t b_cabac;                  /* b_cabac (default 1): entropy coder whose sizes the RD mode decision uses */
    uint32_t inter;                   /* analyse.inter & (I4x4|PSUB16x16|PSUB8x8)                */
    int32_t i_chroma_qp_offset;       /* analyse.i_chroma_qp_offset                              */
    int32_t i_luma_deadzone[2];       /* analyse.i_luma_deadzone {inter,intra} = {21,11}         */
    int32_t i_tscale;                 /* temporal-candidate scale (common/macroblock.c:454); 256
                                         for consecutive P frames, 0 = previous frame was intra  */
    int32_t i_psy_rd;                 /* h->mb.i_psy_rd = FIX8(analyse.f_psy_rd) (encoder.c:515): 256 by default from subme 6 on,
                                         0 below; the caller also lowers i_chroma_qp_offset as encoder.c:520-521 does */
} pcamv_params_t;

/* One macroblock of the pass-1 record: the members of h->info.cache[] (common/common.h:585-603) that this path produces, under
 * their names and with their meaning; the ORDER of the members differs and the reference's (anonymous) struct carries more
 * (intra fields), so a host copies member by member (integration/pcamv_x264_glue.c).  mv[] / ref[] use x264's block-index
 * order 0 1 4 5 / 2 3 6 7 / 8 9 12 13 / 10 11 14 15 (analyse.c:2893-2898). */
typedef struct pcamv_mb_t {
    int32_t i_type;                   /* PCAMV_P_L0 | PCAMV_P_8x8 | PCAMV_P_SKIP                 */
    int32_t i_partition;              /* PCAMV_D_16x16 | 16x8 | 8x16 | 8x8                       */
    int32_t i_qp;
    uint8_t i_sub_partition[4];
    int8_t  ref[16];
    int16_t mv[16][2];
    int16_t mv_stego[16][2];
    int32_t inter_stego_cost[16];
    int16_t pskip_mv[2];              /* pskip_mv_ (encoder.c:1265)                              */
    int16_t mvr16[2];                 /* h->mb.mvr[0][0][mb]: the 16x16 search result            */
    uint8_t used;
    uint8_t pad[3];
} pcamv_mb_t;

/* Per-frame embedding vectors (h->info.cover/rho_final/message/stego/filp, common.h:604-617).
 * The caller allocates each array with room for 16 * mb_count entries. */
typedef struct pcamv_embed_t {
    int32_t  n;                       /* number of carrier MVs (h->info.length)                  */
    int32_t  m;                       /* number of message bits embedded                         */
    int32_t  stc_ok;                  /* stc_embed return value (ignored by the reference)       */
    int32_t  num_flip;
    uint8_t *cover;
    float   *rho;
    uint8_t *message;
    uint8_t *stego;
    int8_t  *flip;
} pcamv_embed_t;

typedef struct pcamv_ctx pcamv_ctx_t;
typedef struct pcamv_batch pcamv_batch_t;

int  pcamv_gpu_open(const pcamv_params_t *param, int device, pcamv_ctx_t **ctx);
void pcamv_gpu_close(pcamv_ctx_t *ctx);
const char *pcamv_gpu_last_error(const pcamv_ctx_t *ctx);

/* Source picture, I420, caller-owned host planes. */
int pcamv_gpu_upload_fenc(pcamv_ctx_t *ctx, const uint8_t *const plane[3], const int stride[3]);

/* Reconstructed reference picture (already deblocked by the caller), I420 host planes, plus
 * the previous frame's final motion field for the temporal candidates
 * (x264_mb_predict_mv_ref16x16, common/macroblock.c:444-467): prev_mv is [mb_h*4][mb_w*4][2]
 * quarter-pel, prev_ref is [mb_h*2][mb_w*2]; both NULL when the reference is an I frame.
 * Produces the padded full/H/V/HV luma planes and padded chroma on the device. */
int pcamv_gpu_set_ref(pcamv_ctx_t *ctx, const uint8_t *const plane[3], const int stride[3],
                      const int16_t *prev_mv, const int8_t *prev_ref);

/* Read back the 4 padded luma planes produced by set_ref: out must hold 4*stride*lines
 * bytes; *stride = ALIGN(width+64,16), *lines = height+64, origin at (32,32).  The caller gets x264's
 * raster planes (filtered[0..3] with their borders); on the device they are kept in a strip layout
 * (DESIGN.md section 3), which this call undoes. */
int pcamv_gpu_get_ref_planes(pcamv_ctx_t *ctx, uint8_t *out, int *stride, int *lines);

/* Pass-1 analysis of the whole P frame at luma QP qp.  out_mb[mb_count] receives the record
 * (with embed != 0 also used / mv_stego / inter_stego_cost).  recon[3] (optional, tightly
 * packed w*h, w/2*h/2 x2) receives the pass-1 reconstruction before deblocking. */
int pcamv_gpu_analyse_pframe(pcamv_ctx_t *ctx, int qp, int embed, pcamv_mb_t *out_mb,
                             uint8_t *const recon[3]);

/* Embedding stage on the records of the last analyse call.  emrate as x264's --emrate
 * (encoder.c:1828-1836): 0 < r <= 1 bits per MV, r > 1 bits per frame.  message == NULL
 * draws the bits from the context's glibc-compatible rand() stream (seed 1 at open, as the
 * reference never calls srand, encoder.c:1838-1840); otherwise message[0..m) is used. */
int pcamv_gpu_embed_pframe(pcamv_ctx_t *ctx, float emrate, const uint8_t *message, int message_len,
                           pcamv_embed_t *out);

/* Apply the flip map to the record (pass-2 substitution, analyse.c:3001-3107): final MVs. */
int pcamv_gpu_final_mvs(pcamv_ctx_t *ctx, pcamv_mb_t *out_mb);

/* Pass 2 of the frame last analysed (analyse.c:2870-3107 + x264_macroblock_encode, then the loop filter
 * x264_frame_deblock_row, common/frame.c:627-798): final MVs = the record with mv_stego where flips[k] == 1
 * (k = carrier index in embedding order; flips == NULL: the flip map the last embed_pframe left on the
 * device), P_SKIP macroblocks take the skip prediction from their final neighbours; out_final (optional)
 * receives the record with the final MVs, recon[3] the reconstruction before the loop filter, deblocked[3]
 * after it.  The deblocked picture stays in the context's reconstruction planes (the next reference) and the
 * final motion field is what PCAMV_PREV_FIELD_INTERNAL hands to the next frame.  One QP per frame (CQP). */
int pcamv_gpu_pass2_pframe(pcamv_ctx_t *ctx, const uint8_t *flips, int n_flips, pcamv_mb_t *out_final,
                           uint8_t *const recon[3], uint8_t *const deblocked[3]);

/* Syndrome-trellis extraction (host side of the BER check): stego bits -> message bits (stc_extract, embed.h:340-393).
 * Sub-matrix widths 2..20 come from the code's tables; outside that range (payloads below 1/20 bit per MV, or 1 bit per MV) the
 * reference draws the columns from a process-wide LCG (embed.h:134-139) that every such frame advances, so the extractor needs
 * the generator's state: *lcg = state before this frame's embedding (1 for the first P frame of a process -- of a closed GOP
 * under the per-GOP parity definition; a context starts there), updated to the state after it, to be carried to the next frame
 * exactly as the reference's own extractor process carries it.  pcamv_gpu_stc_extract = the same with lcg == NULL: tabulated
 * widths only, PCAMV_EUNSUP otherwise. */
int pcamv_gpu_stc_extract(const uint8_t *stego, int n, int m, int matrixheight, uint8_t *message);
int pcamv_gpu_stc_extract_lcg(const uint8_t *stego, int n, int m, int matrixheight, int64_t *lcg, uint8_t *message);

/* H.264 MV-syntax extractor, the decode side of the BER check (the reference has none): parses the slice data
 * of a CABAC-coded P slice -- the bytes from the first macroblock's mb_skip_flag on, i.e. after the slice header and its
 * cabac_alignment bits, as encoder/cabac.c writes them (arithmetic decoding engine, mb_skip_flag, mb_type, sub_mb_type, mvd,
 * coded_block_pattern, mb_qp_delta, the residual, end_of_slice_flag) -- and returns every macroblock's i_type, i_partition,
 * i_sub_partition, ref[] and mv[] (MV prediction of H.264 8.4.1; P_SKIP inferred) in the record's layout; the other members are 0.
 * Host code, no GPU needed.  Scope: what the P slices of this path contain (frame macroblocks, one reference so no ref_idx,
 * 4x4 transform, cabac_init_idc 0): PCAMV_EUNSUP for an intra macroblock, PCAMV_EINVAL for a stream that does not end with
 * the picture's last macroblock.  The carrier LSBs of the result go to pcamv_gpu_stc_extract*. */
int pcamv_gpu_parse_pslice_cabac(const uint8_t *slice_data, size_t len, int mb_w, int mb_h, int slice_qp, pcamv_mb_t *out_mb);
/* the same for a CAVLC-coded P slice (--no-cabac): slice_data = from the first mb_skip_run to the rbsp trailing bits, as
 * encoder/cavlc.c writes them (mb_skip_run, mb_type, sub_mb_type, mvd, coded_block_pattern, mb_qp_delta, residual_block_cavlc) */
int pcamv_gpu_parse_pslice_cavlc(const uint8_t *slice_data, size_t len, int mb_w, int mb_h, pcamv_mb_t *out_mb);
/* The same on what a stream really holds.  pcamv_gpu_nal_to_rbsp undoes x264_nal_encode (common/common.c:658-690): optional Annex-B
 * start code, the NAL header byte (returned), emulation_prevention_three_bytes removed; rbsp must hold len bytes.  The _at forms
 * start at bit start_bit of the RBSP, i.e. right behind a slice header of any length (the caller parses or skips the header:
 * it depends on the SPS / PPS in use): CAVLC slice data begins at that bit, CABAC slice data after the cabac_alignment_one_bits
 * that fill up the byte. */
int pcamv_gpu_nal_to_rbsp(const uint8_t *nal, size_t len, uint8_t *rbsp, size_t *rbsp_len, int *nal_ref_idc, int *nal_unit_type);
int pcamv_gpu_parse_pslice_cabac_at(const uint8_t *rbsp, size_t len, size_t start_bit, int mb_w, int mb_h, int slice_qp, pcamv_mb_t *out_mb);
int pcamv_gpu_parse_pslice_cavlc_at(const uint8_t *rbsp, size_t len, size_t start_bit, int mb_w, int mb_h, pcamv_mb_t *out_mb);

/* Device-resident variants used by bench.py and the multi-frame pipeline: planes are raw
 * device pointers (hipMalloc / torch storage), tightly packed like recon[] above. */
int pcamv_gpu_set_ref_device(pcamv_ctx_t *ctx, const void *y, const void *u, const void *v,
                             const void *prev_mv, const void *prev_ref);
/* pass as prev_mv and prev_ref to chain frames on the device: the temporal candidates then come
 * from the motion field the context's previous step produced (kept in a ping-pong buffer) */
#define PCAMV_PREV_FIELD_INTERNAL ((const void *)(uintptr_t)1)
int pcamv_gpu_set_fenc_device(pcamv_ctx_t *ctx, const void *y, const void *u, const void *v);
/* One full step on resident inputs: plane production + analysis + RCA + embedding; results
 * stay on the device until pcamv_gpu_fetch_results. stream = hipStream_t as void*. */
int pcamv_gpu_step_device(pcamv_ctx_t *ctx, int qp, float emrate, void *stream);
int pcamv_gpu_fetch_results(pcamv_ctx_t *ctx, pcamv_mb_t *out_mb, pcamv_embed_t *out);
/* Batches: several contexts (independent closed GOPs of the same size on the same device) advance
 * one frame per step together; every kernel launch then carries the same dependency step of all
 * of them.  A single 1080p frame exposes at most 60 independent macroblocks at a time,
 * the batch is what fills the 256 CUs.  The contexts keep their own inputs/outputs (set_ref_device,
 * set_fenc_device, fetch_results); the batch only owns the launch descriptors and the scheduler
 * state; destroy a batch before closing its contexts.
 * Scheduling of the analysis inside a step: by default ONE persistent launch ("k_analyse_flow") in
 * which ready macroblocks of all GOPs flow through a device queue (search -> publish motion ->
 * RCA costs -> reconstruction per macroblock); with PCAMV_SCHED=diag in the environment at
 * batch/context creation, one launch per anti-diagonal ("k_search_diag") followed by RCA and
 * reconstruction launches.  Results are identical; kernel_time takes the name of the active one.
 * The closed loop's second pass (pass 2 + loop filter) follows the same choice: one persistent launch
 * ("k_pass2_deblock_flow") through the same queue, or one launch per anti-diagonal. */
int  pcamv_gpu_batch_create(pcamv_ctx_t *const *ctxs, int n, pcamv_batch_t **batch);
void pcamv_gpu_batch_destroy(pcamv_batch_t *batch);
int  pcamv_gpu_batch_step(pcamv_batch_t *batch, int qp, float emrate, void *stream);
int  pcamv_gpu_batch_kernel_time(pcamv_batch_t *batch, const char *kernel, double *avg_ms, int *launches, int reset);
const char *pcamv_gpu_batch_last_error(const pcamv_batch_t *batch);
/* Closed loop on the device: with on != 0 every batch step ends with pass 2 + the loop filter (embedding must be
 * on: the flip map comes from it), leaving each context's deblocked reconstruction in its device planes
 * (pcamv_gpu_recon_device) -- hand those to set_ref_device as the next frame's reference. */
int  pcamv_gpu_batch_set_closed_loop(pcamv_batch_t *batch, int on);
int  pcamv_gpu_recon_device(pcamv_ctx_t *ctx, void *planes[3]);
/* host copy of those planes (tightly packed w*h, w/2*h/2 x2) after synchronising */
int  pcamv_gpu_fetch_recon(pcamv_ctx_t *ctx, uint8_t *const planes[3]);
/* name of the analysis kernel the batch's schedule launches ("k_analyse_flow", "k_analyse_flow_rd" with --subme >= 6,
 * "k_analyse_flow_tesa", or "k_search_diag") */
const char *pcamv_gpu_batch_dominant_kernel(const pcamv_batch_t *batch);
/* Results of the step enqueued last, for every context of the batch, copied on `stream` without synchronising the host:
 * context i's pass-1 records (n_mb x pcamv_mb_t) to (char *)dst_mb + i * mb_stride and, when dst_flip != NULL, its flip map
 * (16 * n_mb bytes, the first `n carriers` meaningful) to (char *)dst_flip + i * flip_stride.  The destinations may be device
 * memory or pinned host memory; the caller orders `stream` after the step's stream (an event) and owns the synchronisation.
 * This is what lets a host pipeline overlap a step's downloads with the next step (bench.py pcie_inclusive). */
int  pcamv_gpu_batch_copy_results_async(pcamv_batch_t *batch, void *dst_mb, size_t mb_stride, void *dst_flip, size_t flip_stride, void *stream);

/* Average duration in ms of the dominant kernel over the launches since the last reset,
 * measured with hipEvents on the launch stream (bench.py roofline). */
int pcamv_gpu_kernel_time(pcamv_ctx_t *ctx, const char *kernel, double *avg_ms, int *launches, int reset);

/* Pixel-metric probe for parity tests of the cost functions themselves (common/pixel.c SAD/SATD,
 * common/mc.c get_ref / mc_chroma): n requests {mb_x, mb_y, i_pixel, xoff, yoff, mvx, mvy (qpel),
 * satd}, each answered with {luma cost, U cost, V cost} of the uploaded fenc block against the
 * current reference at that MV (no MV-bit cost added). */
int pcamv_gpu_block_costs(pcamv_ctx_t *ctx, int qp, int n, const int32_t *req, int32_t *out);

/* Probe of the RD stage's metrics and intra predictors on caller-supplied pixels, for parity tests against reference-minted vectors
 * (common/pixel.c:71-96 ssd, :256-358 sa8d / hadamard_ac; encoder/rdo.c:106-137 ssd_mb with the psy-RD term; encoder/analyse.c:522-549
 * x264_mb_cache_fenc_satd; common/predict.c 16x16 / chroma 8x8 / 4x4 predictors scored with satd -- sad at subme 1 -- as the intra
 * analysis of analyse.c:552-879 does).  n requests of 1024 bytes: source macroblock (Y 16x16 rows of 16, then 8 rows of U | V),
 * a second macroblock in the same layout, intra borders top[3][28] ([c][3] = top-left sample, [c][4 + x]; luma 24 wide) and
 * left[3][16], int32 avail at byte 900 (bit 0 left, bit 1 top).  out: 32 int32 per request -- 0 ssd 16x16 + 2 x 8x8 of source vs
 * second block, 1 the same + the psy term (ssd_mb), 2 / 3 hadamard_ac 16x16 of the second block's luma (4x4 / 8x8 energies),
 * 4 / 5 fenc_satd_sum / fenc_sa8d_sum of the source, 6..9 intra 16x16 V, H, DC (variant by avail), P, 10..13 intra chroma DC
 * (variant), H, V, P over both planes, 14..25 the twelve 4x4 modes on block 0 (x264's I_PRED_4x4 order); unavailable = 1 << 28. */
int pcamv_gpu_rd_probe(pcamv_ctx_t *ctx, int qp, int n, const uint8_t *req, int32_t *out);

/* Diagnostics: record every block-cost evaluation {ip,xoff,yoff,mvx,mvy,flags,cost,cost2} made for
 * macroblock mb by the next analyse call (mb < 0: off); out holds 1 + 8*4000 int32, out[0] = count. */
int pcamv_gpu_trace_mb(pcamv_ctx_t *ctx, int mb);
int pcamv_gpu_trace_fetch(pcamv_ctx_t *ctx, int32_t *out);

/* Diagnostics for the parity tests of the RD mode decision (--subme >= 6 with CABAC): FNV-1a of the 460 context states
 * after every macroblock of the following analyses; out holds mb_count words. */
int pcamv_gpu_debug_state_hash(pcamv_ctx_t *ctx, int enable);
int pcamv_gpu_debug_state_hash_fetch(pcamv_ctx_t *ctx, uint32_t *out);

int pcamv_gpu_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
