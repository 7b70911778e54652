/*
 * pcamv_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar) of the reference's pass-1 P-frame analysis and
 * embedding stage.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libpcamv_gpu.so) never links or calls it.
 *
 * Pinning: checked against the reference's own C code compiled by oracle/Makefile
 * (oracle/_ref/libpcamv_ref.so) through tests/golden/ fixtures minted by
 * oracle/gen_golden.py -- the reference ships no golden vectors of its own.
 * Not pinned (documented in DESIGN.md): the cover/cost assembly of encoder.c:1561-1855,
 * whose translation unit cannot be built here.
 */
#ifndef PCAMV_ORACLE_H
#define PCAMV_ORACLE_H
#include <stdint.h>
#include "../include/pcamv_gpu.h"

typedef struct orc orc_t;

orc_t *orc_open(const pcamv_params_t *p);
void   orc_close(orc_t *o);

/* planes: tightly packed I420 */
void orc_set_fenc(orc_t *o, const uint8_t *y, const uint8_t *u, const uint8_t *v);
void orc_set_ref(orc_t *o, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                 const int16_t *prev_mv, const int8_t *prev_ref);
int  orc_ref_stride(const orc_t *o);
int  orc_ref_lines(const orc_t *o);
void orc_get_ref_planes(const orc_t *o, uint8_t *out4);
void orc_get_ref_integral(const orc_t *o, uint16_t *out);

int  orc_analyse_pframe(orc_t *o, int qp, int embed, pcamv_mb_t *out_mb,
                        uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v);
/* diagnostics: FNV-1a of the 460 CABAC context states after every macroblock (subme >= 6, b_cabac) */
void orc_set_debug(orc_t *o, uint32_t *state_hash, int dump_mb, uint8_t *dump_state /* [460] or NULL */);
int  orc_embed_pframe(orc_t *o, const pcamv_mb_t *mbs, float emrate, const uint8_t *message, int message_len,
                      pcamv_embed_t *out);
int orc_pass2_pframe(orc_t *o, int qp, const pcamv_mb_t *mbs, const uint8_t *flips, int n_flips, pcamv_mb_t *out, uint8_t *nnz,
                     uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint8_t *dbk_y, uint8_t *dbk_u, uint8_t *dbk_v);
void orc_final_mvs(const orc_t *o, const pcamv_embed_t *e, pcamv_mb_t *mbs);

int  orc_stc_embed(const uint8_t *cover, int n, const uint8_t *msg, int m, const float *rho,
                   uint8_t *stego, int matrixheight);
void orc_stc_lcg_reset(long state);
int  orc_stc_extract(const uint8_t *stego, int n, int m, int matrixheight, uint8_t *msg);

/* glibc-compatible rand() (TYPE_3 additive feedback), for the message stream */
typedef struct { int32_t r[34]; int f, b; } orc_rand_t;
void orc_srand(orc_rand_t *s, unsigned seed);
int  orc_rand(orc_rand_t *s);

/* primitives, for checkasm-style differential tests */
int  orc_sad(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb);
int  orc_satd(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb);
int  orc_ssd(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb);
int  orc_sa8d(int i_pixel, const uint8_t *a, int sa, const uint8_t *b, int sb);              /* PIX_16x16, PIX_8x8 */
uint64_t orc_hadamard_ac(int i_pixel, const uint8_t *pix, int stride);                        /* PIX_16x16 .. PIX_8x8 */
void orc_predict(int kind, int mode, uint8_t *dst /* stride 32, neighbours in place */);     /* 0: 16x16, 1: chroma 8x8, 2: 4x4 */
void orc_cabac_init_p(uint8_t *state /* [460] */, int qp);
void orc_mc_luma(uint8_t *dst, int ds, uint8_t *const src[4], int ss, int mvx, int mvy, int w, int h);
void orc_mc_chroma(uint8_t *dst, int ds, const uint8_t *src, int ss, int mvx, int mvy, int w, int h);
void orc_cost_mv_table(int qp, int16_t *out /* [4*4*2048+1] */);
void orc_me_search(orc_t *o, int qp, int mb_x, int mb_y, int i_pixel, int xoff, int yoff,
                   const int16_t mvp[2], const int16_t (*mvc)[2], int i_mvc, int16_t out_mv[2], int out_cost[2]);

#endif
