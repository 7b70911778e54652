/*
 * pcamv_common.h -- types and tables shared by the HIP kernels and their host launcher.
 *
 * Data layout in HBM (one set per context):
 *   fenc Y/U/V      tightly packed w*h, (w/2)*(h/2) x2
 *   reference luma  4 planes (full, H, V, HV half-pel) of stride = ALIGN16(w+64), h+64 lines, the
 *                   picture origin at (32,32): the same values as x264's filtered[0..3] planes
 *                   (common/frame.c:46-77) after border expansion
 *   reference chroma 2 planes, stride = ALIGN16(w/2+32), h/2+32 lines, origin (16,16)
 *   motion field    mv[mb_h*4][mb_w*4][2] int16 qpel, ref8[mb_h*2][mb_w*2], mb_type[n_mb],
 *                   mvr[n_mb][2] (16x16 search results): h->mb.mv/ref/type/mvr (common.h:433-444)
 *   record          pcamv_mb_t[n_mb] (include/pcamv_gpu.h) + mvp_aux[n_mb][16][2]
 */
#ifndef PCAMV_COMMON_H
#define PCAMV_COMMON_H
#include <stddef.h>
#include <stdint.h>
#include "../../include/pcamv_gpu.h"

#ifdef PCAMV_HOST_EMU
#define PCAMV_DEV static inline
#define PCAMV_CONST static const
#else
#include <hip/hip_runtime.h>
#define PCAMV_DEV __device__ __forceinline__
#define PCAMV_CONST __device__ static const
#endif

#define PCAMV_PAD 32
#define PCAMV_CPAD 16
/* Luma reference planes in HBM are stored as vertical STRIPS: strip s holds the padded-plane columns 28 s .. 28 s + 31 (its last
 * four are the next strip's first four again), row after row with a pitch of 32 bytes, strips one after the other:
 *     byte(x, y) = s * 32 * lines + y * 32 + (x - 28 s),   s = x / 28.
 * A 128-byte cache line is 32 x 4 pixels instead of 128 x 1: the 48 x 48 window of a search is ~34 lines instead of ~66, the
 * 24 x 20 neighbourhood of a sub-pel refinement ~10 instead of ~24 per plane.  Thanks to the four repeated columns an (unaligned)
 * 4-byte fetch at any x never straddles two strips, a row step is +32 whatever the picture size (scalar row bases), and x + 1
 * needs no second look-up.  Writer: k_hpel (groups of 4 pixels are aligned with the strips: 28 = 7 x 4).  Chroma planes stay raster. */
#define PCAMV_LSW 28
#define PCAMV_LROW 32
#define PCAMV_LSTRIPS(stride) (((stride) + PCAMV_LSW - 1) / PCAMV_LSW + 1)
#define PCAMV_LSTRIP_OF(x) (((uint32_t)(x) * 18725u) >> 19)          /* x / 28 for x < 40000 (device code: v_mul_u32_u24, see lsw_x) */
#define PCAMV_COST_MAX (1 << 28)
#define PCAMV_COST_MV_LEN (4 * 4 * 2048 + 1)
#define PCAMV_COST_MV_CENTRE (2 * 4 * 2048)
#define SCAN8_0 (4 + 1 * 8)

enum { PIX_16x16, PIX_16x8, PIX_8x16, PIX_8x8, PIX_8x4, PIX_4x8, PIX_4x4 };

/* everything a kernel needs to know about one frame; passed by value */
struct FrameDev {
    int w, h, mb_w, mb_h, n_mb;
    int stride, lines, cstride, clines;
    long long plane_size;          /* strips * 32 * lines: the four luma planes are contiguous, plane k = luma_base + k*plane_size */
    int lskip;                     /* 32 * lines - 28: byte(x, y) = y * 32 + x + (x / 28) * lskip */
    long long cplane_size;         /* the two chroma planes are contiguous too: chroma_base[1] = chroma_base[0] + cplane_size */
    const uint8_t *fenc[3];
    const uint8_t *raw[3];         /* un-padded reference planes the plane-production kernels read */
    uint8_t *luma_base, *chroma_base[2];   /* start of the padded allocations */
    uint8_t *luma_raster;          /* --me esa / tesa only: the full-pel plane once more in raster rows (their row primitives walk x-consecutive positions: 8-byte row loads shared by four candidates), else NULL */
    uint8_t *luma[4];              /* CPU emulation of the control code only (tests/emu): picture-origin pointers into RASTER padded planes */
    uint8_t *chroma[2];
    uint8_t *rec[3];               /* pass-1 reconstruction out, tightly packed */
    int8_t *mb_type;
    int16_t *mv;                   /* [n_mb*16][2] */
    int8_t *ref8;
    int16_t *mvr;                  /* [n_mb][2] */
    const int16_t *prev_mv;
    const int8_t *prev_ref;
    int have_prev, tscale;
    pcamv_mb_t *rec_mb;            /* the pass-1 record */
    int16_t *mvp_aux;              /* [n_mb][16][2] search-time mvp of each carrier slot */
    const int16_t *cost_mv;        /* centre pointer of the lambda*bits table for this QP */
    /* parameters */
    int qp, chroma_qp, lambda, chroma_qp_offset;
    int me_method, me_range, subme, mv_range, b_chroma_me, b_fast_pskip, b_dct_decimate, b_cabac;
    unsigned inter;
    int embed;
    /* quantiser (flat matrices): 3 position classes */
    int q_mf[2][3], q_bias[2][3], dq_mf[3];    /* [0] luma inter, [1] chroma inter; at qp / chroma_qp */
    int dq_mf_c[3];
    int lambda2_chroma;            /* x264_lambda2_tab[chroma_qp] for the skip-probe SSD threshold */
    int *trace; int trace_mb;      /* diagnostics: log every block-cost evaluation of one MB (trace[0] = count) */
    /* pass 2 (final MVs -> reconstruction -> loop filter) */
    uint16_t *nnz;                 /* [n_mb] bit i: luma 4x4 block i (x264 block order) kept non-zero levels */
    const int8_t *flip;            /* flip map of the embedding stage, one entry per carrier MV in embedding order */
    const int *car_base;           /* [n_mb] index of the macroblock's first carrier in that order */
    /* --subme >= 6: RD mode decision (encoder/rdo.c).  What the entropy coder's contexts of the right / lower neighbours
     * read from a coded macroblock, and the slice's CABAC context states handed from macroblock to macroblock */
    int b_mbrd, psy_rd, lambda2, ref_is_inter;
    int q_mf_i[3], q_bias_i[3];    /* intra luma quantiser (CQM_4IY) at qp, for the 4x4 intra analysis */
    uint8_t *nb_nz;                /* [n_mb][16] non-zero flags (CABAC) / coefficient counts (CAVLC): 0..7 bottom row (4 luma, 2 Cb, 2 Cr), 8..15 right column */
    int16_t *nb_cbp;               /* [n_mb] h->mb.cbp: luma | chroma << 4 | chroma DC coded bits << 9 */
    int16_t *nb_mvd;               /* [n_mb][8][2] MV differences of the bottom row (0..3) and right column (4..7) of 4x4 blocks */
    uint8_t *cabac;                /* [PCAMV_CHAIN_BYTES] what the macroblock coded last hands to the next one in coding order: the context
                                    * states [0, 464) (460 used) and, with sub-8x8 partitions, its own non-zero flags / counts and MV
                                    * differences (x264's cache keeps them and x264_rd_cost_part reads them: PCAMV_CHAIN_NZ / _MVD) */
    const uint8_t *cabac_init;     /* [PCAMV_CHAIN_BYTES] the same at the slice start: states for this QP (H.264 9.3.1.1, cabac_init_idc 0), zeros */
    const uint32_t *cabac_tab;     /* [256] per (state, bin): 8.8 fixed-point bits << 8 | next state */
    uint32_t *dbg_hash;            /* diagnostics: [n_mb] FNV-1a of the context states after each macroblock, or NULL */
    int rec_is_pass1;              /* pass 2: rec / nnz still hold what this frame's first pass left (nothing has filtered them yet) */
    const uint8_t *mbflip;         /* pass 2: [n_mb] 1 = some carrier of the macroblock is flipped in `flip` (k_mb_flips), or NULL: look at the carriers */
};

/* Small lookup tables live in registers as packed constants: a table in memory costs one global
 * load round trip per use, and these sit on the serial chain of the search (me.c tables, block
 * geometry of common/macroblock.c).  4 bits per entry, biased where entries are signed. */
PCAMV_DEV int nib32(unsigned k, int i) { return (int)((k >> (4 * i)) & 15u); }
PCAMV_DEV int nib64(unsigned long long k, int i) { return (int)((k >> (4 * i)) & 15ull); }
PCAMV_DEV int lg_w4_of(int ip) { return nib32(0x0011122u, ip); }          /* log2(width / 4):  16,16,8,8,8,4,4 */
PCAMV_DEV int lg_h4_of(int ip) { return nib32(0x0101212u, ip); }          /* log2(height / 4): 16,8,16,8,4,8,4 */
PCAMV_DEV int lg_nblk_of(int ip) { return nib32(0x0112334u, ip); }        /* log2 of 4x4 blocks per partition */
PCAMV_DEV int pix_w_of(int ip) { return 4 << lg_w4_of(ip); }
PCAMV_DEV int pix_h_of(int ip) { return 4 << lg_h4_of(ip); }
PCAMV_DEV int blk_x_of(int idx) { return (idx & 1) | ((idx >> 1) & 2); }  /* x264 block order -> 4x4 column / row */
PCAMV_DEV int blk_y_of(int idx) { return ((idx >> 1) & 1) | ((idx >> 2) & 2); }
PCAMV_DEV int scan8_of(int idx) { return SCAN8_0 + blk_x_of(idx) + 8 * blk_y_of(idx); }
PCAMV_DEV int clip3i(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
PCAMV_DEV int iabs(int v) { return v < 0 ? -v : v; }
PCAMV_DEV int imin(int a, int b) { return a < b ? a : b; }
PCAMV_DEV int imax(int a, int b) { return a > b ? a : b; }
PCAMV_DEV int median3i(int a, int b, int c) { return imax(imin(a, b), imin(imax(a, b), c)); }
PCAMV_DEV int size_ue_of(unsigned v) { int n = 0; for (v++; v > 1; v >>= 1) n++; return 2 * n + 1; }   /* Exp-Golomb code length (common/bs.h:174-273) */
/* intra 4x4 prediction modes (common/predict.h:76-92) */
enum { I4_V, I4_H, I4_DC, I4_DDL, I4_DDR, I4_VR, I4_HD, I4_VL, I4_HU, I4_DC_LEFT, I4_DC_TOP, I4_DC_128 };

/* Per-macroblock working set.  On the GPU this lives in LDS (one wavefront = one macroblock);
 * every lane executes the control code redundantly on wave-uniform values. */
struct MBLocal {
    /* (round 3: what the second pass of a macroblock touches comes first -- up to and including `pred` --, so that its kernel only
     * allocates that much LDS, PCAMV_PASS2_LDS, and fits six waves per SIMD instead of four; recb .. the union stay contiguous in this
     * order: the Hadamard exhaustive search's survivor list runs across them, TESA_SLOT) */
    uint8_t fenc[24 * 16];         /* Y 16x16 then U|V 8x8 side by side, stride 16 (x264 fenc_buf layout) */
    int16_t cmv[48][2];
    int8_t cref[48];
    int16_t pskip_mv[2];
    int16_t mvr_own[2];            /* this macroblock's 16x16 search result as published to F.mvr */
    int mb_x, mb_y, mb_xy;
    int mv_min[2], mv_max[2], mv_min_spel[2], mv_max_spel[2], mv_min_fpel[2], mv_max_fpel[2];
    int neighbour, type_left, type_top, type_topleft, type_topright;
    int i_type, i_partition;
    uint8_t sub_part[4];
    int b_skip_mc, cbp_luma, cbp_chroma;
    int nnz_mask;                  /* luma blocks with non-zero levels after the transform stage (bit = x264 block index) */
    uint32_t cxy[64];              /* candidate list of the running evaluation: x | y << 16, quarter-pel; CAND_NONE = skip */
    int ccost[192];                /* cost of every listed candidate ([64..191]: per-plane chroma terms of the probe kernel) */
    int slots[16];
    uint8_t recb[24 * 16];         /* reconstruction in the same layout (fenc_buf_ih) */
    uint8_t recb0[24 * 16];        /* reconstruction of the macroblock as decided: shared by all its carriers' first RCA step */
    uint8_t pred[24 * 16];         /* prediction -> reconstruction, same layout */
    union {
        struct {                       /* P_SKIP probe (search phase): per 4x4 block levels and their summaries */
            int16_t coef[24][16];      /* luma 0..15, U 16..19, V 20..23 */
            int16_t cdc[2][4];
            int blk_nz[24], blk_score[24];
            int red[64];               /* scratch for cross-lane work */
        };
        uint8_t pred4[4][24 * 16];     /* RCA phase: prediction -> reconstruction of four re-encodes made together */
    };
    int mvc16[9][2];               /* candidate MVs of the 16x16 search */
    int nbc[12];                   /* neighbourhood costs of the RCA step */
    /* reference window of the RCA step of a 16x16 macroblock: every MV it touches lies within +-3 quarter
     * pels of the decided one, so 4 luma planes of 24 x 20 and 2 chroma planes of 16 x 12 bytes hold all
     * the reference pixels of its re-encodes and nine-point lists */
    uint32_t win[(4 * 480 + 2 * 192) / 4];
    int win_x0, win_y0, win_cx0, win_cy0;     /* plane coordinates (padded-plane origin) of the window's first byte */
    /* --subme >= 6 */
    uint8_t ib_top[3][28];         /* intra prediction neighbours (unfiltered pass-1 reconstruction): the line above, [c][3] = top left, [c][4 + x] */
    uint8_t ib_left[3][16];        /* ... and the column to the left */
    uint8_t nzc[48];               /* h->mb.cache.non_zero_count (scan8 layout; 0x80 = unavailable) */
    int16_t cmvd[48][2];           /* h->mb.cache.mvd */
    int8_t i4mode[48];             /* h->mb.cache.intra4x4_pred_mode */
    int cbp_left, cbp_top;         /* h->mb.cache.i_cbp_left / top, -1 = unavailable */
    int b_fast_intra, fenc_satd_sum, fenc_sa8d_sum;
    /* the RD trial that is the best so far, kept whole: what encoding + entropy-coding the macroblock as decided would produce
     * again (reconstruction, non-zero flags / counts, MV differences, coded block pattern; the context states it ends in
     * are kept in the window storage, L_CABK) */
    uint8_t snap_pred[24 * 16];
    uint8_t snap_nzc[48];
    int16_t snap_cmvd[48][2];
    int snap_cbp_luma, snap_cbp_chroma, snap_nnz_mask;
    int snap_part, snap_cost;      /* partition (PCAMV_D_*) of the kept trial, -1 = none; its RD cost */
#ifdef PCAMV_SEARCH_CALL           /* the motion search as a callee (pcamv_logic.h): its arguments pass through here */
    const struct FrameDev *fdesc;  /* this macroblock's frame descriptor in the descriptor array (constant memory) */
    int me_tmp[10];                /* MEState */
    int mvc_tmp[9][2];
#endif
};
#define PCAMV_PASS2_LDS ((int)offsetof(MBLocal, coef))       /* everything up to and including pred */
/* the second-pass kernel allocates PCAMV_PASS2_LDS bytes of an MBLocal: what it touches must lie below the cut, and two users count on
 * neighbours staying neighbours (the residual walk's rows run from cxy into ccost; the Hadamard exhaustive search's survivor list and the
 * intra analysis' picture run from recb on) */
static_assert(offsetof(MBLocal, pred) + sizeof(((MBLocal *)0)->pred) == offsetof(MBLocal, coef), "pred must end at the second pass' LDS cut");
static_assert(offsetof(MBLocal, fenc) < PCAMV_PASS2_LDS && offsetof(MBLocal, cmv) < PCAMV_PASS2_LDS && offsetof(MBLocal, cref) < PCAMV_PASS2_LDS &&
              offsetof(MBLocal, nnz_mask) < PCAMV_PASS2_LDS && offsetof(MBLocal, slots) + sizeof(((MBLocal *)0)->slots) <= PCAMV_PASS2_LDS &&
              offsetof(MBLocal, ccost) + sizeof(((MBLocal *)0)->ccost) <= PCAMV_PASS2_LDS, "a field the second pass uses lies beyond its LDS");
static_assert(offsetof(MBLocal, ccost) == offsetof(MBLocal, cxy) + sizeof(((MBLocal *)0)->cxy), "cxy and ccost must be contiguous (cab_residual_walk)");
static_assert(offsetof(MBLocal, recb0) == offsetof(MBLocal, recb) + 384 && offsetof(MBLocal, pred) == offsetof(MBLocal, recb) + 768 &&
              offsetof(MBLocal, coef) == offsetof(MBLocal, recb) + 1152, "recb, recb0, pred, the union must be contiguous in this order (TESA_SLOT, L_IFD)");
/* Storage that is idle while the RD decision runs is reused (no LDS growth): the RCA reference window holds the CABAC
 * context states (slice states, a trial copy of the macroblock-header contexts) and the (bits, next state) table; the RCA
 * reconstruction buffers hold the intra 4x4 analysis' picture (17 rows of 32: row -1 and column -1 are the neighbours);
 * the P_SKIP probe's coefficient scratch (coef / cdc) holds a trial's quantised levels in scan order. */
#define PCAMV_CHAIN_NZ 464          /* 24 bytes: non_zero_count of blocks 0..23 */
#define PCAMV_CHAIN_MVD 488         /* 16 x 4 bytes: mvd of the luma blocks 0..15 (x264 block order) */
#define PCAMV_CHAIN_BYTES 576
#define L_CAB(L, k) ((uint8_t *)(L)->win + 464 * (k))      /* k = 0: the slice's states as the macroblock coded last left them */
#define PCAMV_CAB_USED 276                                  /* contexts 0..275 are all a P slice of this path touches (coeff_abs_level_minus1 ends at 275) */
#define L_CABT(L) ((uint8_t *)(L)->win + 464)               /* states at the end of the running size trial */
#define L_CABK(L) ((uint8_t *)(L)->win + 464 + PCAMV_CAB_USED)  /* ... of the trial kept as the best so far */
#define L_CTAB(L) ((L)->win + 256)
#define L_IFD(L) ((uint8_t *)(L)->recb)
#define IFD(L, x, y) (L_IFD(L)[((y) + 1) * 32 + (x) + 4])
/* common/common.h:217-238: cache position of block idx (0..15 luma, 16..19 Cb, 20..23 Cr, 24 luma DC, 25 Cb DC, 26 Cr DC) */
PCAMV_DEV int scan8_all_of(int idx)
{
    if (idx < 16) return scan8_of(idx);
    if (idx < 24) return 1 + ((idx - 16) & 1) + 8 * (1 + (((idx - 16) >> 1) & 1) + 3 * ((idx - 16) >> 2));
    return 4 + (idx - 24) + 5 * 8;
}

/* TESA (me.c:525-600): the positions that survive the ADS / SAD thresholds, one word each: (SAD + MV bits) << 12 |
 * window row << 6 | window column.  Up to 32 x 33 of them (me_range <= 16); they live in LDS that is idle while a
 * search runs: the reconstruction buffers + transform scratch (RCA / probe only), then the RCA window. */
#define TESA_MAX_RANGE 16
#define TESA_REGION_A ((int)((3 * 24 * 16 + 4 * 24 * 16) / 4))
#define TESA_SLOT(L, i) ((i) < TESA_REGION_A ? (uint32_t *)((uint8_t *)(L) + offsetof(MBLocal, recb)) + (i) : (L)->win + ((i) - TESA_REGION_A))
#define TESA_PACK(sad, ry, rx) ((uint32_t)(sad) << 12 | (uint32_t)(ry) << 6 | (uint32_t)(rx))
#define TESA_SAD(e) ((int)((e) >> 12))
#define NB_LEFT 1
#define NB_TOP 2
#define NB_TOPRIGHT 4
#define NB_TOPLEFT 8

/* Candidate lists.  The search code writes up to 64 quarter-pel candidates into L->cxy and asks
 * prim_eval_list for all their costs at once (pixel metric + MV bits [+ chroma]); the primitive
 * leaves every cost in L->ccost and returns the smallest one with the FIRST index reaching it --
 * exactly what the reference's sequential `if (cost < bcost)` over the same candidates in the same
 * order produces (COST_MV / COST_MV_X4 of encoder/me.c, COST_MV_SATD, MV_SATD_FDEC_IH). */
#define CAND_PACK(X, Y) ((uint32_t)(uint16_t)(X) | ((uint32_t)(uint16_t)(Y) << 16))
#define CAND_NONE 0x80008000u
#define CAND_X(c) ((int)(int16_t)(L->cxy[c] & 0xffffu))
#define CAND_Y(c) ((int)(int16_t)(L->cxy[c] >> 16))
#define EV_SATD 1      /* 4x4 Hadamard metric (x264 satd, 8x4-pair rounding) instead of SAD */
#define EV_CHROMA 2    /* add the U and V cost of the co-located chroma block (partitions >= 8x8) */
#define EV_FPEL 4      /* promise: every candidate is full-pel (single-plane fetch) */
#define EV_NOMV 8      /* do not add the MV bit cost */
#define EV_PROBE 16    /* chroma terms go to ccost[64 + c] (U) and ccost[128 + c] (V) instead of being added */
#define EV_WIN 32      /* every candidate lies in the LDS reference window (L->win): no global loads */
#define EV_SRC4 64     /* candidate c is measured against source block enc + (c & 3) * 384 (four re-encodes side by side; 16x16 only) */
#define WIN_LW 24      /* luma window: bytes per row, rows, bytes per plane */
#define WIN_LH 20
#define WIN_LP (WIN_LW * WIN_LH)
#define WIN_CW 16      /* chroma window */
#define WIN_CH 12
#define WIN_CP (WIN_CW * WIN_CH)
struct EvalRes { int cost, idx; };
/* lane-parallel generation of a candidate list: the body runs once per candidate index c < n (n <= 64) */
#ifdef PCAMV_HOST_EMU
#define FOR_CAND(c, n) for (int c = 0; c < (n); c++)
#define NB_SLOT(i) (i)          /* where iteration i of a FOR_CAND loop keeps a value for a later FOR_CAND loop: element i on the CPU */
#else
#define FOR_CAND(c, n) for (int c = LANE(), c##_1 = 1; c##_1 && c < (n); c##_1 = 0)
#define NB_SLOT(i) 0           /* ... the lane's own register on the GPU (the body runs once per lane) */
#endif

/* Stores of the bytes other wavefronts read inside the same launch (a macroblock's final motion, read by its right
 * and lower neighbours): write-through at agent scope (`global_store ... sc1`), so that publishing them needs only the
 * storing wave's `s_waitcnt vmcnt(0)` and not an agent-scope release, which writes back the XCD's whole dirty L2
 * (guide 6 Guideline 16, recipe R1; the consumer keeps its agent-scope acquire). */
#ifdef PCAMV_HOST_EMU
#define NB_ST32(p, v) (*(uint32_t *)(p) = (uint32_t)(v))
#define NB_ST16(p, v) (*(uint16_t *)(p) = (uint16_t)(v))
#define NB_ST8(p, v) (*(int8_t *)(p) = (int8_t)(v))
#define NB_LD32(p) (*(const uint32_t *)(p))
#define NB_LD8(p) (*(const int8_t *)(p))
#define NB_LD16(p) (*(const uint16_t *)(p))
#else
#define NB_ST32(p, v) __hip_atomic_store((uint32_t *)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define NB_ST16(p, v) __hip_atomic_store((uint16_t *)(p), (uint16_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define NB_ST8(p, v) __hip_atomic_store((int8_t *)(p), (int8_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define NB_LD32(p) __hip_atomic_load((uint32_t *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define NB_LD8(p) __hip_atomic_load((int8_t *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define NB_LD16(p) __hip_atomic_load((uint16_t *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#define NB_PACK16(lo, hi) ((uint32_t)(uint16_t)(lo) | (uint32_t)(uint16_t)(hi) << 16)

#ifdef PCAMV_HOST_EMU
#define PCAMV_WAVE_SYNC() do { } while (0)
#else
/* Lanes of the wavefront exchange data through LDS (one lane writes, another reads).  The hardware
 * executes a wave's LDS operations in order, but to the compiler these are unsynchronised accesses of
 * different threads which it may reorder (it does: a lane-0 store was forwarded to lane 0 while the
 * other lanes' load of the same word was moved in front of it).  A wavefront-scope release/acquire
 * pair emits no instruction and pins the order; every primitive begins and ends with one. */
#define PCAMV_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

struct MEState {
    int i_pixel, xoff, yoff;
    int mvp[2];
    int cost_mv, cost, cost_rec;
    int mv[2];
};


/* diagnostics build (-DPCAMV_PROF): wave cycles per phase of k_analyse_flow, summed in pcamv_prof[] (tools/dbg/prof_phases.py) */
#if defined(PCAMV_PROF) && !defined(PCAMV_HOST_EMU)
#define PCAMV_PROF_N 48
static __device__ unsigned long long pcamv_prof[PCAMV_PROF_N];
/* a wave (= a workgroup of the kernels that are timed) sums in LDS and adds to the global counters once, when it leaves: with an
 * atomic per event on one address, 4096 waves spent their time on exactly that (a profile 20x slower than the kernel) */
__shared__ unsigned long long pcamv_prof_l[PCAMV_PROF_N];
#define PROF_T() __builtin_readcyclecounter()
#define PROF_ADD(i, t0) do { if ((threadIdx.x & 63) == 0) pcamv_prof_l[i] += (unsigned long long)(PROF_T() - (t0)); } while (0)
#define PROF_CNT(i, n) do { if ((threadIdx.x & 63) == 0) pcamv_prof_l[i] += (unsigned long long)(n); } while (0)
#define PROF_INIT() do { if (threadIdx.x < PCAMV_PROF_N) pcamv_prof_l[threadIdx.x] = 0; __syncthreads(); } while (0)
#define PROF_FLUSH() do { __syncthreads(); if (threadIdx.x < PCAMV_PROF_N && pcamv_prof_l[threadIdx.x]) atomicAdd(&pcamv_prof[threadIdx.x], pcamv_prof_l[threadIdx.x]); } while (0)
#else
#define PROF_CNT(i, n) do { } while (0)
#define PROF_T() 0ull
#define PROF_ADD(i, t0) do { (void)(t0); } while (0)
#define PROF_INIT() do { } while (0)
#define PROF_FLUSH() do { } while (0)
#endif
#endif
