/*
 * pcamv_prims_gpu.h -- lane-parallel pixel work for one wavefront (64 lanes) = one macroblock.
 *
 * Lane layout for block costs: a candidate block of w x h pixels is cut into its
 * nblk = (w/4)*(h/4) 4x4 sub-blocks; lane = cand * nblk + blk, so one wavefront evaluates
 * 64/nblk >= 4 candidates at once.  Each lane loads its 4x4 pixels (4 unaligned dword loads
 * per plane, v_lerp_u8 for the quarter-pel average), computes SAD (v_sad_u8) or the 4x4
 * Hadamard in registers, and the per-candidate total is a DPP butterfly over nblk lanes
 * (quad_perm / row_half_mirror / row_mirror: no LDS traffic), read back with v_readlane.
 * The reference frame is read straight from HBM/L2 with coalesced-enough dword loads; the
 * source macroblock, prediction and reconstruction live in LDS (MBLocal).
 *
 * Arithmetic restated from the reference: common/pixel.c:40-65,187-253 (SAD/SATD),
 * common/mc.c:194-277 (get_ref/mc_luma/mc_chroma), common/dct.c:122-232, common/quant.c:33-109,
 * 203-239 (decimate), encoder/macroblock.c:71-85 (dct2x2dc).
 */
#ifndef PCAMV_PRIMS_GPU_H
#define PCAMV_PRIMS_GPU_H
#include "pcamv_common.h"

/* The lane number passes through an empty asm wherever it is asked for: the compiler then cannot move what is computed from it
 * (lane == k flags, lane-derived offsets: dozens of values) out of the persistent kernel's macroblock loop, where they lived for
 * the whole launch -- in scratch, reloaded at ~800 sites of the RD instance, every reload a memory round trip of its own in front
 * of the instruction that needs it.  Recomputing them where they are used is one or two VALU instructions. */
__device__ __forceinline__ int pcamv_lane_id(void) { int l = (int)(threadIdx.x & 63); asm volatile("" : "+v"(l)); return l; }
#ifdef PCAMV_RD_LO      /* the one-wave-per-SIMD build has the registers to keep them (183 in use): hoisted is 1-2 % faster there */
#define LANE() ((int)(threadIdx.x & 63))
#else
#define LANE() pcamv_lane_id()
#endif
/* clamp to [0,255] of an already shifted value.  The empty asm keeps hipcc (ROCm 7.2) from fusing
 * shift + clamp of two neighbours into v_ashr_pk_u8_i32: the code it emits around that gfx950
 * instruction ORs further bytes into the destination assuming bits [31:16] come back zero, but
 * the hardware leaves the old contents there (seen as wrong bytes 2/3 of every packed dword: first in k_hpel, in round 3 by the
 * predictor probe in the plane predictors of the intra analysis, whose values can leave [0,255] before the clamp). */
__device__ __forceinline__ uint32_t clamp_u8(int v)
{
    asm volatile("" : "+v"(v));
    return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}
__device__ __forceinline__ uint32_t ld4u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t lds4(const uint8_t *p) { return *(const uint32_t *)p; }
__device__ __forceinline__ void sts4(uint8_t *p, uint32_t v) { *(uint32_t *)p = v; }
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
__device__ __forceinline__ int dpp_qp1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_qp2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_hmir(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_mir(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); }

/* sum over aligned groups of g lanes (g = 1,2,4,8,16); every lane of a group gets the total */
__device__ __forceinline__ int group_sum(int v, int g)
{
    if (g >= 2) v += dpp_qp1(v);
    if (g >= 4) v += dpp_qp2(v);
    if (g >= 8) v += dpp_hmir(v);
    if (g >= 16) v += dpp_mir(v);
    return v;
}
__device__ __forceinline__ int wave_sum_all(int v)
{
    v = group_sum(v, 16);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ int hadamard4x4_abs(const uint32_t e[4], const uint32_t r[4])
{
    int t[4][4], s = 0;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        int d0 = (int)(e[y] & 255) - (int)(r[y] & 255), d1 = (int)((e[y] >> 8) & 255) - (int)((r[y] >> 8) & 255);
        int d2 = (int)((e[y] >> 16) & 255) - (int)((r[y] >> 16) & 255), d3 = (int)(e[y] >> 24) - (int)(r[y] >> 24);
        int s01 = d0 + d1, d01 = d0 - d1, s23 = d2 + d3, d23 = d2 - d3;
        t[y][0] = s01 + s23; t[y][1] = d01 + d23; t[y][2] = s01 - s23; t[y][3] = d01 - d23;
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
        int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
        s += iabs(s01 + s23) + iabs(d01 + d23) + iabs(s01 - s23) + iabs(d01 - d23);
    }
    return s;
}

/* 4 chroma pixels of mc_chroma (mc.c:246-277) at chroma-plane position (cx,cy) */
__device__ __forceinline__ uint32_t chroma_row4(const FrameDev &F, int plane, int cx, int cy, int mvx, int mvy)
{
    int dx = mvx & 7, dy = mvy & 7;
    int cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
    const uint8_t *s = (plane ? F.chroma[1] : F.chroma[0]) + (ptrdiff_t)(cy + (mvy >> 3)) * F.cstride + cx + (mvx >> 3);
    uint32_t a0 = ld4u(s), a1 = s[4], b0 = ld4u(s + F.cstride), b1 = s[F.cstride + 4];
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int p00 = (a0 >> (8 * i)) & 255, p01 = i < 3 ? (a0 >> (8 * i + 8)) & 255 : a1;
        int p10 = (b0 >> (8 * i)) & 255, p11 = i < 3 ? (b0 >> (8 * i + 8)) & 255 : b1;
        o |= (uint32_t)((cA * p00 + cB * p01 + cC * p10 + cD * p11 + 32) >> 6) << (8 * i);
    }
    return o;
}


__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = imin(v, dpp_qp1(v)); v = imin(v, dpp_qp2(v)); v = imin(v, dpp_hmir(v)); v = imin(v, dpp_mir(v));
    return imin(imin(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                imin(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

/* global-memory accessors: a wave-uniform base (scalar registers) + a 32-bit per-lane byte offset, so
 * the loads are `global_load_* v, voffset, s[base]` with no 64-bit vector address arithmetic */
typedef const __attribute__((address_space(1))) uint8_t *gp8;
typedef const __attribute__((address_space(1))) int16_t *gp16;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gld4(gp8 base, uint32_t off) { return *(const __attribute__((address_space(1))) u32u *)(base + off); }
__device__ __forceinline__ uint64_t gld8(gp8 base, uint32_t off) { return *(const __attribute__((address_space(1))) u64u *)(base + off); }
/* entry i of the MV-bit table (int16): scalar table base + a 32-bit byte offset (a pointer + index form makes the compiler do
 * 64-bit vector address arithmetic in every list pass) */
__device__ __forceinline__ int gld_cost(gp8 tab, uint32_t i) { return (int)*(const __attribute__((address_space(1))) int16_t *)(tab + (i << 1)); }
/* one wave-uniform entry of the MV-bit table for the control code (the cost of the search's current best MV, of its start point): a
 * SCALAR load (236 cycles on average under load against 580 for a vector load; the table is constant during a launch).  d = mv - mvp,
 * relative to the table's centre; the containing dword is fetched and the half picked */
__device__ __forceinline__ int prim_mv_cost(const FrameDev &F, int d)
{
    const int i = rfl(d) + PCAMV_COST_MV_CENTRE;
    const __attribute__((address_space(4))) uint32_t *t = (const __attribute__((address_space(4))) uint32_t *)(F.cost_mv - PCAMV_COST_MV_CENTRE);
    const uint32_t w = t[i >> 1];
    return (int)(int16_t)(w >> (16 * (i & 1)));
}
/* luma planes are strips (pcamv_common.h): the x part of a pixel's byte offset; the y part is y * PCAMV_LROW.  4 bytes from there
 * (and from x + 1: the repeated columns) lie in one strip */
/* 32-bit integer multiplies run at a quarter of the VALU rate on this chip; where both factors are known to stay below 2^24
 * (pixel coordinates, strides, quantiser factors, bilinear weights, levels) the 24-bit forms give the same low 32 bits at full rate */
__device__ __forceinline__ uint32_t mul24u(uint32_t a, uint32_t b) { return __umul24(a, b); }
__device__ __forceinline__ int mul24s(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ int mad24s(int a, int b, int c) { return __mul24(a, b) + c; }
__device__ __forceinline__ uint32_t lsw_x(uint32_t x, uint32_t lskip) { return mul24u(mul24u(x, 18725u) >> 19, lskip) + x; }
__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }

/* four rows of 4 pixels -> per column x the pair (row0[x], row1[x]) in o[x] and (row2[x], row3[x]) in
 * o[4 + x], zero-extended to 16 bits: the layout the packed Hadamard works on */
__device__ __forceinline__ void pk_cols(const uint32_t r[4], uint32_t o[8])
{
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t sel = 0x0c040c00u + (uint32_t)x * 0x00010001u;
        o[x] = __builtin_amdgcn_perm(r[1], r[0], sel);
        o[4 + x] = __builtin_amdgcn_perm(r[3], r[2], sel);
    }
}
/* HALF the sum of |4x4 Hadamard of (e - r)| in packed 16-bit arithmetic (two rows per register).
 * The last butterfly is folded with |a+b| + |a-b| = 2 max(|a|,|b|), so the full sum is even and
 * x264's satd rounding ((sum_a [+ sum_b]) >> 1, pixel.c:187-253) is exact: it IS this half. */
__device__ __forceinline__ int satd4x4_half(const uint32_t ec[8], const uint32_t r[4])
{
    uint32_t rc[8];
    pk_cols(r, rc);
    v2s t[8];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        v2s d0 = as_v2s(ec[4 * g]) - as_v2s(rc[4 * g]), d1 = as_v2s(ec[4 * g + 1]) - as_v2s(rc[4 * g + 1]);
        v2s d2 = as_v2s(ec[4 * g + 2]) - as_v2s(rc[4 * g + 2]), d3 = as_v2s(ec[4 * g + 3]) - as_v2s(rc[4 * g + 3]);
        v2s s01 = d0 + d1, d01 = d0 - d1, s23 = d2 + d3, d23 = d2 - d3;
        t[4 * g] = s01 + s23; t[4 * g + 1] = d01 + d23; t[4 * g + 2] = s01 - s23; t[4 * g + 3] = d01 - d23;
    }
    v2s acc = {0, 0};
    const v2s z = {0, 0};
#pragma unroll
    for (int x = 0; x < 4; x++) {
        v2s S = t[x] + t[4 + x], D = t[x] - t[4 + x];
        v2s aS = __builtin_elementwise_max(S, z - S), aD = __builtin_elementwise_max(D, z - D);
        v2s U = as_v2s(__builtin_amdgcn_perm(as_u32(aD), as_u32(aS), 0x05040100u));   /* (|S.lo|, |D.lo|) */
        v2s V = as_v2s(__builtin_amdgcn_perm(as_u32(aD), as_u32(aS), 0x07060302u));   /* (|S.hi|, |D.hi|) */
        acc += __builtin_elementwise_max(U, V);
    }
    return (int)acc.x + (int)acc.y;
}

/* mc_chroma (mc.c:246-277) of a 4x4 block: rows as 8-byte loads, the bilinear weights as one
 * v_dot4_u32_u8 per output pixel over the bytes {A[i], A[i+1], B[i], B[i+1]} */
__device__ __forceinline__ void chroma_rows_load(gp8 cb, uint32_t cstride, uint32_t off, uint64_t row[5])
{
#pragma unroll
    for (int k = 0; k < 5; k++) row[k] = gld8(cb + (size_t)k * cstride, off);
}
__device__ __forceinline__ void chroma_block4_rows(const uint64_t row[5], int mvx, int mvy, uint32_t r[4])
{
    const int dx = mvx & 7, dy = mvy & 7;
    const uint32_t W = mul24u(8 - dx, 8 - dy) | mul24u(dx, 8 - dy) << 8 | mul24u(8 - dx, dy) << 16 | mul24u(dx, dy) << 24;
    uint32_t w[5][4];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t lo = (uint32_t)row[k], hi = (uint32_t)(row[k] >> 32);
        w[k][0] = lo; w[k][1] = __builtin_amdgcn_alignbit(hi, lo, 8); w[k][2] = __builtin_amdgcn_alignbit(hi, lo, 16); w[k][3] = __builtin_amdgcn_alignbit(hi, lo, 24);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t px = __builtin_amdgcn_perm(w[k + 1][i], w[k][i], 0x05040100u);
            o |= (__builtin_amdgcn_udot4(px, W, 32u, false) >> 6) << (8 * i);
        }
        r[k] = o;
    }
}
__device__ __forceinline__ void chroma_block4(gp8 cb, uint32_t cstride, uint32_t off, int mvx, int mvy, uint32_t r[4])
{
    uint64_t row[5];
    chroma_rows_load(cb, cstride, off, row);
    chroma_block4_rows(row, mvx, mvy, r);
}

/* ---- LDS reference window (RCA of a 16x16 macroblock) ---- */
/* 4 bytes at byte offset b of the window: two aligned words + v_alignbit (LDS wants natural alignment) */
__device__ __forceinline__ uint32_t wld4(const MBLocal *L, int b)
{
    const uint32_t *w = L->win + (b >> 2);
    return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)(b & 3) * 8u);
}
/* load the window around the decided MV (bmx,bmy): luma planes from (x0,y0) = MB origin + ((mv - 3) >> 2),
 * chroma from ((mv - 3) >> 3); coalesced row segments, one dword per lane and step */
__device__ __forceinline__ void prim_win_load(const FrameDev &F, MBLocal *L, int bmx, int bmy)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    const int x0 = L->mb_x * 16 + PCAMV_PAD + ((bmx - 3) >> 2), y0 = L->mb_y * 16 + PCAMV_PAD + ((bmy - 3) >> 2);
    const int cx0 = L->mb_x * 8 + PCAMV_CPAD + ((bmx - 3) >> 3), cy0 = L->mb_y * 8 + PCAMV_CPAD + ((bmy - 3) >> 3);
    const gp8 lb = (gp8)F.luma_base, cb = (gp8)F.chroma_base[0];
    const uint32_t lskip = (uint32_t)F.lskip, psz = (uint32_t)F.plane_size, cstride = (uint32_t)F.cstride, cps = (uint32_t)F.cplane_size;
    for (int i = lane; i < 4 * WIN_LH * (WIN_LW / 4); i += 64) {
        const int pl = i / (WIN_LH * (WIN_LW / 4)), r = (i / (WIN_LW / 4)) % WIN_LH, c = i % (WIN_LW / 4);
        L->win[i] = gld4(lb, (uint32_t)pl * psz + (uint32_t)(y0 + r) * PCAMV_LROW + lsw_x((uint32_t)(x0 + 4 * c), lskip));
    }
    for (int i = lane; i < 2 * WIN_CH * (WIN_CW / 4); i += 64) {
        const int pl = i / (WIN_CH * (WIN_CW / 4)), r = (i / (WIN_CW / 4)) % WIN_CH, c = i % (WIN_CW / 4);
        L->win[4 * WIN_LP / 4 + i] = gld4(cb, (uint32_t)pl * cps + (uint32_t)(cy0 + r) * cstride + (uint32_t)(cx0 + 4 * c));
    }
    if (lane == 0) { L->win_x0 = x0; L->win_y0 = y0; L->win_cx0 = cx0; L->win_cy0 = cy0; }
    PCAMV_WAVE_SYNC();
}
/* chroma_block4 out of the window: b = byte offset of the block's first reference pixel */
__device__ __forceinline__ void chroma_block4_win(const MBLocal *L, int b, int mvx, int mvy, uint32_t r[4])
{
    const int dx = mvx & 7, dy = mvy & 7;
    const uint32_t W = mul24u(8 - dx, 8 - dy) | mul24u(dx, 8 - dy) << 8 | mul24u(8 - dx, dy) << 16 | mul24u(dx, dy) << 24;
    const uint32_t sh = (uint32_t)(b & 3) * 8u;
    uint32_t w[5][4];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t *p = L->win + ((b + k * WIN_CW) >> 2);
        const uint32_t lo = __builtin_amdgcn_alignbit(p[1], p[0], sh), hi = __builtin_amdgcn_alignbit(p[2], p[1], sh);
        w[k][0] = lo; w[k][1] = __builtin_amdgcn_alignbit(hi, lo, 8); w[k][2] = __builtin_amdgcn_alignbit(hi, lo, 16); w[k][3] = __builtin_amdgcn_alignbit(hi, lo, 24);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t px = __builtin_amdgcn_perm(w[k + 1][i], w[k][i], 0x05040100u);
            o |= (__builtin_amdgcn_udot4(px, W, 32u, false) >> 6) << (8 * i);
        }
        r[k] = o;
    }
}

/* Costs of the n <= 64 candidates listed in L->cxy for the block (ip at xoff,yoff) against the source
 * rows in enc (LDS: L->fenc or L->recb).
 *   luma:   lane = slot * nblk + blk, 64/nblk candidates per pass, passes back to back; each lane
 *           fetches its 4x4 reference pixels (one or two planes, v_lerp_u8), SAD (v_sad_u8) or the
 *           packed Hadamard, DPP butterfly over the nblk lanes, MV bits looked up per lane, the
 *           group's first lane stores the total to ccost[c];
 *   chroma: (partitions >= 8x8) lane = slot * 2nb + plane * nb + blk, both planes of 128/nblk
 *           candidates per pass, group totals added to ccost[c] with an LDS atomic;
 *   result: lane c reads ccost[c], key = cost << 6 | c, wave minimum -> smallest cost, first index. */
/* what a list evaluation reads of the frame descriptor */
struct EvalEnv { const int16_t *cost_mv; const uint8_t *luma_base, *chroma_base; long long plane_size, cplane_size; int lskip, cstride; int *trace; int trace_mb; };
__device__ __forceinline__ EvalRes eval_list_body(const EvalEnv &F, MBLocal *L, const uint8_t *enc, int ip_, int xoff_, int yoff_,
                                                  int n_, int flags_, int mvp0_, int mvp1_)
{
    const int ip = rfl(ip_), xoff = rfl(xoff_), yoff = rfl(yoff_), n = rfl(n_), flags = rfl(flags_), mvp0 = rfl(mvp0_), mvp1 = rfl(mvp1_);
    const int lane = LANE();
    const int lgn = lg_nblk_of(ip), lgw = lg_w4_of(ip), nblk = 1 << lgn;
    const int satd = flags & EV_SATD;
    const gp8 cost_tab = (gp8)(F.cost_mv - PCAMV_COST_MV_CENTRE);
    PCAMV_WAVE_SYNC();
    const unsigned long long t_ev = PROF_T();
    PROF_CNT(32, 1); PROF_CNT(33, n);
    int key_acc = 0x7fffffff;
    {
        const int slot = lane >> lgn, blk = lane & (nblk - 1);
        const int px = xoff + 4 * (blk & ((1 << lgw) - 1)), py = yoff + 4 * (blk >> lgw);
        uint32_t e[4], ec[8];
        const uint8_t *encl = (flags & EV_SRC4) ? enc + (slot & 3) * 384 : enc;       /* the slot's source stays the same in every pass */
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = lds4(encl + (py + k) * 16 + px);
        if (satd) pk_cols(e, ec);
        const gp8 lb = (gp8)F.luma_base;
        const uint32_t stride = PCAMV_LROW, psz = (uint32_t)F.plane_size, lskip = (uint32_t)F.lskip;
        const gp8 lb1 = lb + stride, lb2 = lb1 + stride, lb3 = lb2 + stride;
        const uint32_t rowbase = (uint32_t)(L->mb_y * 16 + py + PCAMV_PAD) * stride, colbase = (uint32_t)(L->mb_x * 16 + px + PCAMV_PAD);
        const int cpp = 64 >> lgn;
        for (int p0 = 0; p0 < n; p0 += cpp) {
            const int c = p0 + slot;
            const uint32_t xy = L->cxy[c < n ? c : 0];
            const bool act = c < n && xy != CAND_NONE;
            const int mvx = act ? (int)(int16_t)(xy & 0xffffu) : 0, mvy = act ? (int)(int16_t)(xy >> 16) : 0;
            const uint32_t o = rowbase + (uint32_t)((mvy >> 2) * (int)stride) + lsw_x(colbase + (uint32_t)(mvx >> 2), lskip);
            /* the candidate's MV bits (L1-resident table) are asked for BEFORE its pixels, and both are waited for together: left
             * where it is used, the look-up ends up inside the blk == 0 branch behind the pixels' s_waitcnt -- a second memory
             * round trip in every pass */
            int mvb0 = 0, mvb1 = 0;
            if (!(flags & EV_NOMV)) { mvb0 = gld_cost(cost_tab, (uint32_t)(mvx - mvp0 + PCAMV_COST_MV_CENTRE)); mvb1 = gld_cost(cost_tab, (uint32_t)(mvy - mvp1 + PCAMV_COST_MV_CENTRE)); }
            uint32_t r[4];
            if (flags & EV_WIN) {
                const int dx = mvx & 3, dy = mvy & 3;
                const int wb = (L->mb_y * 16 + py + PCAMV_PAD + (mvy >> 2) - L->win_y0) * WIN_LW + (L->mb_x * 16 + px + PCAMV_PAD + (mvx >> 2) - L->win_x0);
                const int ba = wb + ((dx != 0) + 2 * (dy == 2)) * WIN_LP + (dy == 3 ? WIN_LW : 0);
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = wld4(L, ba + k * WIN_LW);
                if ((dx | dy) & 1) {
                    const int bb = wb + (dy ? (2 + (dx == 2)) * WIN_LP : 0) + (dx == 3);
#pragma unroll
                    for (int k = 0; k < 4; k++) r[k] = avg4(r[k], wld4(L, bb + k * WIN_LW));
                }
            } else if (flags & EV_FPEL) {
                r[0] = gld4(lb, o); r[1] = gld4(lb1, o); r[2] = gld4(lb2, o); r[3] = gld4(lb3, o);
            } else {
                /* get_ref (mc.c:194-243): plane pair by the quarter-pel phase, in arithmetic form of hpel_ref0/1 */
                const int dx = mvx & 3, dy = mvy & 3;
                const uint32_t oa = o + (dx != 0 ? psz : 0u) + (dy == 2 ? 2u * psz : 0u) + (dy == 3 ? stride : 0u);
                r[0] = gld4(lb, oa); r[1] = gld4(lb1, oa); r[2] = gld4(lb2, oa); r[3] = gld4(lb3, oa);
                if ((dx | dy) & 1) {
                    const uint32_t ob = o + (dy ? (dx == 2 ? 3u * psz : 2u * psz) : 0u) + (dx == 3);
                    r[0] = avg4(r[0], gld4(lb, ob)); r[1] = avg4(r[1], gld4(lb1, ob));
                    r[2] = avg4(r[2], gld4(lb2, ob)); r[3] = avg4(r[3], gld4(lb3, ob));
                }
            }
            asm volatile("" : "+v"(mvb0), "+v"(mvb1), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
            int v;
            if (satd) v = satd4x4_half(ec, r);
            else {
                uint32_t sa = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) sa = __builtin_amdgcn_sad_u8(e[k], r[k], sa);
                v = (int)sa;
            }
            v = group_sum(v, nblk) + mvb0 + mvb1;
            if (blk == 0 && c < n) L->ccost[c] = act ? v : PCAMV_COST_MAX;
            if (blk == 0 && act) key_acc = imin(key_acc, (v << 6) | c);
        }
    }
    PCAMV_WAVE_SYNC();
    if ((flags & EV_CHROMA) && ip <= PIX_8x8) {
        const int lgnb = lgn - 2, nb = 1 << lgnb, lgcw = lgw - 1;          /* chroma 4x4 blocks per plane: 4,2,2,1 */
        const int slot = lane >> (lgnb + 1), plane = (lane >> lgnb) & 1, blk = lane & (nb - 1);
        const int px = (xoff >> 1) + 4 * (blk & ((1 << lgcw) - 1)), py = (yoff >> 1) + 4 * (blk >> lgcw);
        uint32_t e[4], ec[8];
        const uint8_t *encl = (flags & EV_SRC4) ? enc + (slot & 3) * 384 : enc;
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = lds4(encl + 256 + (py + k) * 16 + plane * 8 + px);
        if (satd) pk_cols(e, ec);
        const gp8 cb = (gp8)F.chroma_base;
        const uint32_t cstride = (uint32_t)F.cstride;
        const uint32_t rowbase = (plane ? (uint32_t)F.cplane_size : 0u) + mul24u((uint32_t)(L->mb_y * 8 + py + PCAMV_CPAD), cstride) + (uint32_t)(L->mb_x * 8 + px + PCAMV_CPAD);
        const int cpp = 64 >> (lgnb + 1);
        for (int p0 = 0; p0 < n; p0 += cpp) {
            const int c = p0 + slot;
            const uint32_t xy = L->cxy[c < n ? c : 0];
            const bool act = c < n && xy != CAND_NONE;
            const int mvx = act ? (int)(int16_t)(xy & 0xffffu) : 0, mvy = act ? (int)(int16_t)(xy >> 16) : 0;
            uint32_t r[4];
            if (flags & EV_WIN)
                chroma_block4_win(L, 4 * WIN_LP + plane * WIN_CP + (L->mb_y * 8 + py + PCAMV_CPAD + (mvy >> 3) - L->win_cy0) * WIN_CW
                                         + (L->mb_x * 8 + px + PCAMV_CPAD + (mvx >> 3) - L->win_cx0), mvx, mvy, r);
            else
                chroma_block4(cb, cstride, rowbase + (uint32_t)(mul24s(mvy >> 3, (int)cstride) + (mvx >> 3)), mvx, mvy, r);
            int v;
            if (satd) v = satd4x4_half(ec, r);
            else {
                uint32_t sa = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) sa = __builtin_amdgcn_sad_u8(e[k], r[k], sa);
                v = (int)sa;
            }
            v = group_sum(v, nb);
            if (blk == 0 && act) {
                if (flags & EV_PROBE) L->ccost[64 * (1 + plane) + c] = v;
                else atomicAdd(&L->ccost[c], v);
            }
        }
    }
    PCAMV_WAVE_SYNC();
    /* without a chroma term the totals are still in the registers of the lanes that formed them: the smallest (cost, index) comes
     * straight out of those (the costs go to LDS for the control code that reads single ones, but nothing waits for that) */
    int key = 0x7fffffff;
    if ((flags & EV_CHROMA) && ip <= PIX_8x8) { if (lane < n) { int cc = L->ccost[lane]; if (cc < PCAMV_COST_MAX) key = (cc << 6) | lane; } }
    else key = key_acc;
    key = wave_min_i32(key);
    EvalRes res;
    if (key == 0x7fffffff) { res.cost = PCAMV_COST_MAX; res.idx = -1; }
    else { res.cost = key >> 6; res.idx = key & 63; }
    PROF_ADD(38, t_ev);
#ifdef PCAMV_TRACE      /* diagnostics build only: log every candidate of one macroblock (pcamv_gpu_trace_mb) */
    if (F.trace && L->mb_xy == F.trace_mb && lane == 0)
        for (int c = 0; c < n; c++) {
            if (L->cxy[c] == CAND_NONE) continue;
            int k = F.trace[0];
            if (k < 4000) { int *t = F.trace + 1 + 8 * k; t[0] = ip; t[1] = xoff; t[2] = yoff; t[3] = CAND_X(c); t[4] = CAND_Y(c); t[5] = flags | ((enc == L->recb || enc == L->recb0) ? 32 : 0); t[6] = L->ccost[c]; t[7] = c; F.trace[0] = k + 1; }
        }
#endif
    PCAMV_WAVE_SYNC();
    return res;
}

__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return (uint64_t)(uint32_t)rfl((int)(uint32_t)v) | (uint64_t)(uint32_t)rfl((int)(uint32_t)(v >> 32)) << 32; }
#ifdef PCAMV_EVAL_CALL
/* -DPCAMV_EVAL_CALL (off; kept for measurements): the list primitive as ONE real function (flags at run time; the descriptor
 * fields it reads arrive as values and are made scalar again) instead of an inlined copy per call site (7 KB against ~600 KB
 * of the kernel).  Measured slower both ways: a lone wave with every register 356 k against 282 k cycles per macroblock, the
 * 4-waves-per-SIMD build 12.7 against 14.2 M MB/s at 4096 chains (fewer spills, 128 instead of 149, but 40 calls per macroblock
 * each pay the callee-saved registers' round trip through scratch).  The CABAC residual walk, 2 calls per macroblock, is the
 * opposite case (pcamv_prims_rd_gpu.h). */
static __device__ __noinline__ EvalRes eval_list_fn(EvalEnv E, MBLocal *L, const uint8_t *enc, int ip, int xoff, int yoff, int n, int flags, int mvp0, int mvp1)
{
    EvalEnv U;
    U.cost_mv = (const int16_t *)rfl64((uint64_t)E.cost_mv); U.luma_base = (const uint8_t *)rfl64((uint64_t)E.luma_base); U.chroma_base = (const uint8_t *)rfl64((uint64_t)E.chroma_base);
    U.plane_size = (long long)rfl64((uint64_t)E.plane_size); U.cplane_size = (long long)rfl64((uint64_t)E.cplane_size);
    U.lskip = rfl(E.lskip); U.cstride = rfl(E.cstride); U.trace = (int *)rfl64((uint64_t)E.trace); U.trace_mb = rfl(E.trace_mb);
    return eval_list_body(U, L, enc, ip, xoff, yoff, n, flags, mvp0, mvp1);
}
#endif
__device__ __forceinline__ EvalRes prim_eval_list(const FrameDev &F, MBLocal *L, const uint8_t *enc, int ip, int xoff, int yoff, int n, int flags, int mvp0, int mvp1)
{
    EvalEnv E;
    E.cost_mv = F.cost_mv; E.luma_base = F.luma_base; E.chroma_base = F.chroma_base[0]; E.plane_size = F.plane_size; E.cplane_size = F.cplane_size;
    E.lskip = F.lskip; E.cstride = F.cstride; E.trace = F.trace; E.trace_mb = F.trace_mb;
#ifdef PCAMV_EVAL_CALL
    return eval_list_fn(E, L, enc, ip, xoff, yoff, n, flags, mvp0, mvp1);
#else
    return eval_list_body(E, L, enc, ip, xoff, yoff, n, flags, mvp0, mvp1);
#endif
}

/* Exhaustive search window (me.c:489-622 without the ADS skip, see pcamv_logic.h): SAD + MV bits of every
 * full-pel position x in [min_x, min_x + width), y in [min_y, min_y + nrows), first minimum in raster order.
 * lane = slot * nblk + blk: the 64/nblk slots take 64/nblk consecutive rows; along a row each lane walks x in
 * steps of 4, fetching its four 4-pixel rows as 8 bytes once and forming the four shifted candidates with
 * v_alignbit; the per-candidate sums of a block pair travel through the DPP butterfly packed two to a register
 * (a 16x16 SAD is < 2^16).  key = cost << 11 | raster index (index < 2048). */
__device__ __forceinline__ EvalRes prim_esa_window(const FrameDev &F, MBLocal *L, int ip_, int xoff_, int yoff_, int min_x_, int min_y_,
                                                   int width_, int nrows_, int mvp0_, int mvp1_)
{
    const int ip = rfl(ip_), xoff = rfl(xoff_), yoff = rfl(yoff_), min_x = rfl(min_x_), min_y = rfl(min_y_), width = rfl(width_), nrows = rfl(nrows_);
    const int mvp0 = rfl(mvp0_), mvp1 = rfl(mvp1_);
    const int lane = LANE();
    const int lgn = lg_nblk_of(ip), lgw = lg_w4_of(ip), nblk = 1 << lgn;
    const int slot = lane >> lgn, blk = lane & (nblk - 1), nslot = 64 >> lgn;
    const int px = xoff + 4 * (blk & ((1 << lgw) - 1)), py = yoff + 4 * (blk >> lgw);
    const gp8 cost_tab = (gp8)(F.cost_mv - PCAMV_COST_MV_CENTRE);
    PCAMV_WAVE_SYNC();
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = lds4(L->fenc + (py + k) * 16 + px);
    const gp8 lb = (gp8)F.luma_raster;         /* the raster copy of the full-pel plane (FrameDev): x-consecutive positions share their row loads */
    const uint32_t stride = (uint32_t)F.stride;
    const gp8 lb1 = lb + stride, lb2 = lb1 + stride, lb3 = lb2 + stride;
    const uint32_t rowbase = (uint32_t)(L->mb_y * 16 + py + PCAMV_PAD) * stride + (uint32_t)(L->mb_x * 16 + px + PCAMV_PAD);
    int best = 0x7fffffff;
    for (int r0 = 0; r0 < nrows; r0 += nslot) {
        const int ry = r0 + slot;
        const bool rowok = ry < nrows;
        const int my = min_y + (rowok ? ry : 0);
        const int ycost = gld_cost(cost_tab, (uint32_t)(my * 4 - mvp1 + PCAMV_COST_MV_CENTRE));
        const uint32_t orow = rowbase + (uint32_t)(my * (int)stride + min_x);
        for (int x0 = 0; x0 < width; x0 += 4) {
            const uint32_t o = orow + (uint32_t)x0;
            const uint64_t w0 = gld8(lb, o), w1 = gld8(lb1, o), w2 = gld8(lb2, o), w3 = gld8(lb3, o);
            uint32_t s[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t sh = 8u * (uint32_t)j;
                uint32_t a = 0;
                a = __builtin_amdgcn_sad_u8(e[0], j ? __builtin_amdgcn_alignbit((uint32_t)(w0 >> 32), (uint32_t)w0, sh) : (uint32_t)w0, a);
                a = __builtin_amdgcn_sad_u8(e[1], j ? __builtin_amdgcn_alignbit((uint32_t)(w1 >> 32), (uint32_t)w1, sh) : (uint32_t)w1, a);
                a = __builtin_amdgcn_sad_u8(e[2], j ? __builtin_amdgcn_alignbit((uint32_t)(w2 >> 32), (uint32_t)w2, sh) : (uint32_t)w2, a);
                a = __builtin_amdgcn_sad_u8(e[3], j ? __builtin_amdgcn_alignbit((uint32_t)(w3 >> 32), (uint32_t)w3, sh) : (uint32_t)w3, a);
                s[j] = a;
            }
            int p01 = (int)(s[0] | s[1] << 16), p23 = (int)(s[2] | s[3] << 16);
            p01 = group_sum(p01, nblk); p23 = group_sum(p23, nblk);
            /* lane j < 4 of the group finishes candidate x0 + j (the others repeat one of them: harmless for a
             * minimum); groups of fewer than 4 lanes finish all four in every lane */
            const int jn = nblk >= 4 ? 1 : 4;
            for (int jj = 0; jj < jn; jj++) {
                const int j = nblk >= 4 ? (blk & 3) : jj;
                const int sad = (j & 2 ? p23 : p01) >> (16 * (j & 1)) & 0xffff;
                const int mx = min_x + x0 + j;
                const int cost = sad + ycost + gld_cost(cost_tab, (uint32_t)(mx * 4 - mvp0 + PCAMV_COST_MV_CENTRE));
                const int key = (cost << 11) | (ry * width + x0 + j);
                if (rowok && x0 + j < width && key < best) best = key;
            }
        }
    }
    best = wave_min_i32(best);
    EvalRes res;
    if (best == 0x7fffffff) { res.cost = PCAMV_COST_MAX; res.idx = -1; }
    else { res.cost = best >> 11; res.idx = best & 2047; }
    PCAMV_WAVE_SYNC();
    return res;
}

/* TESA, one row of the window (me.c:539-566): for every full-pel position x in [min_x, min_x + width), y = my:
 * SAD + MV bits of x - min_x (sic) -> L->ccost[x], ADS + x MV bits -> L->ccost[64 + x].  ADS (pixel.c:515-559) = sum over the
 * partition's 8x8 (4x4 below 8x8) sub-blocks of |sum(fenc sub-block) - sum(reference sub-block)|; the reference gets
 * the sums from an integral image, here they come from the pixels the SAD reads anyway.  lane = x (width <= 64). */
__device__ __forceinline__ void prim_tesa_row(const FrameDev &F, MBLocal *L, int ip_, int xoff_, int yoff_, int min_x_, int my_, int width_, int mvp0_)
{
    const int ip = rfl(ip_), xoff = rfl(xoff_), yoff = rfl(yoff_), min_x = rfl(min_x_), my = rfl(my_), width = rfl(width_), mvp0 = rfl(mvp0_);
    const int lane = LANE();
    const int bw = pix_w_of(ip), bh = pix_h_of(ip), sub = ip <= PIX_8x8 ? 8 : 4;
    const gp8 cost_tab = (gp8)(F.cost_mv - PCAMV_COST_MV_CENTRE);
    const gp8 lb = (gp8)F.luma_raster;         /* raster copy of the full-pel plane, as in prim_esa_window */
    const uint32_t stride = (uint32_t)F.stride;
    PCAMV_WAVE_SYNC();
    if (lane < width) {
        const uint32_t base = (uint32_t)(L->mb_y * 16 + yoff + my + PCAMV_PAD) * stride + (uint32_t)(L->mb_x * 16 + xoff + min_x + lane + PCAMV_PAD);
        int sad = 0, rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0, es0 = 0, es1 = 0, es2 = 0, es3 = 0;
        for (int r = 0; r < bh; r++)
            for (int j = 0; j < bw; j += 4) {
                const uint32_t ref = gld4(lb, base + (uint32_t)r * stride + (uint32_t)j), e = lds4(L->fenc + (yoff + r) * 16 + xoff + j);
                sad = (int)__builtin_amdgcn_sad_u8(ref, e, (uint32_t)sad);
                const int rsum = (int)__builtin_amdgcn_sad_u8(ref, 0u, 0u), esum = (int)__builtin_amdgcn_sad_u8(e, 0u, 0u);
                const int k = (r >= sub ? 2 : 0) + (j >= sub ? 1 : 0);
                rs0 += k == 0 ? rsum : 0; rs1 += k == 1 ? rsum : 0; rs2 += k == 2 ? rsum : 0; rs3 += k == 3 ? rsum : 0;
                es0 += k == 0 ? esum : 0; es1 += k == 1 ? esum : 0; es2 += k == 2 ? esum : 0; es3 += k == 3 ? esum : 0;
            }
        /* sub-blocks the partition does not have keep 0 - 0 */
        const int ads = iabs(es0 - rs0) + iabs(es1 - rs1) + iabs(es2 - rs2) + iabs(es3 - rs3);
        /* me.c:551,563: the SAD of a position is charged cost_fpel_mvx[x] with x RELATIVE to the window (the ADS gets the
         * position's real MV bits, cost_fpel_mvx + min_x) -- the reference's arithmetic, kept */
        L->ccost[lane] = sad + gld_cost(cost_tab, (uint32_t)(lane * 4 - mvp0 + PCAMV_COST_MV_CENTRE));
        L->ccost[64 + lane] = ads + gld_cost(cost_tab, (uint32_t)((min_x + lane) * 4 - mvp0 + PCAMV_COST_MV_CENTRE));
    }
    PCAMV_WAVE_SYNC();
}
/* TESA, the reference's walk over that row (me.c:549-566) for all positions at once: a position is looked at when its
 * ADS is below bsad * 17 / 16 (bsad as at the start of the row); walking them in order, one is kept when its SAD is
 * below bsad * sad_thresh >> 3 and lowers bsad when below it.  bsad before position x is therefore
 * min(bsad at row start, smallest looked-at SAD left of x): an exclusive prefix minimum.  Kept positions are appended
 * to the list in order.  Returns the row's final bsad. */
__device__ __forceinline__ int prim_tesa_scan(MBLocal *L, int width_, int bsad_, int sad_thresh_, int ycost_, int ry_, int *n_)
{
    const int width = rfl(width_), bsad0 = rfl(bsad_), sad_thresh = rfl(sad_thresh_), ycost = rfl(ycost_), ry = rfl(ry_), n = rfl(*n_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    const bool look = lane < width && L->ccost[64 + lane] < bsad0 * 17 / 16;
    const int sad = lane < width ? L->ccost[lane] : 0;
    int incl = look ? sad : 0x7fffffff;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl = imin(incl, o); }
    int before = __shfl_up(incl, 1);
    before = lane == 0 ? bsad0 : imin(bsad0, before);
    const bool keep = look && sad < (before * sad_thresh >> 3);
    const unsigned long long m = __ballot(keep);
    if (keep) *TESA_SLOT(L, n + __popcll(m & ((1ull << lane) - 1))) = TESA_PACK(sad + ycost, ry, lane);
    *n_ = n + __popcll(m);
    const int bsad = imin(bsad0, rfl(__shfl(incl, 63)));
    PCAMV_WAVE_SYNC();
    return bsad;
}
/* TESA, pruning of the list (me.c:567-600): above 2 * limit entries keep those with SAD <= bsad * (sad_thresh + 8) >> 4
 * (stable); above limit entries a partial selection sort brings the `limit` smallest to the front (first smallest
 * wins, found entry swapped with the one in its place -- the swaps decide later ties, so they are made as written).
 * The survivors become the candidate list L->cxy.  Returns their number. */
__device__ __forceinline__ int prim_tesa_select(MBLocal *L, int n_, int limit_, int bsad_, int sad_thresh_, int min_x_, int min_y_)
{
    int n = rfl(n_);
    const int limit = rfl(limit_), min_x = rfl(min_x_), min_y = rfl(min_y_);
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    if (n > limit * 2) {
        const int thr = rfl(bsad_) * (rfl(sad_thresh_) + 8) >> 4;
        int kept = 0;
        for (int b0 = 0; b0 < n; b0 += 64) {            /* in place: an entry only ever moves towards the front */
            const uint32_t e = b0 + lane < n ? *TESA_SLOT(L, b0 + lane) : 0u;
            const bool keep = b0 + lane < n && TESA_SAD(e) <= thr;
            const unsigned long long m = __ballot(keep);
            PCAMV_WAVE_SYNC();
            if (keep) *TESA_SLOT(L, kept + __popcll(m & ((1ull << lane) - 1))) = e;
            kept += __popcll(m);
            PCAMV_WAVE_SYNC();
        }
        n = kept;
    }
    if (n > limit) {
        for (int i = 0; i < limit; i++) {
            int best = 0x7fffffff;                      /* SAD << 11 | index: smallest SAD, first index (n <= 1056) */
            for (int j = i + lane; j < n; j += 64) best = imin(best, (TESA_SAD(*TESA_SLOT(L, j)) << 11) | j);
            best = wave_min_i32(best);
            const int bj = best & 2047;
            if (lane == 0 && bj > i) { const uint32_t t = *TESA_SLOT(L, i); *TESA_SLOT(L, i) = *TESA_SLOT(L, bj); *TESA_SLOT(L, bj) = t; }
            PCAMV_WAVE_SYNC();
        }
        n = limit;
    }
    if (lane < n) { const uint32_t e = *TESA_SLOT(L, lane); L->cxy[lane] = CAND_PACK((min_x + (int)(e & 63)) * 4, (min_y + (int)(e >> 6 & 63)) * 4); }
    PCAMV_WAVE_SYNC();
    return n;
}

/* analyse.c:1535-1567: chroma cost of one 8x8 split below 8x8; mv4[k] = MV of luma 4x4 k (raster in the 8x8) */
__device__ __forceinline__ int prim_chroma4x4_cost(const FrameDev &F, MBLocal *L, int i8, const int mv4x[4], const int mv4y[4], int satd)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    int v = 0;
    if (lane < 2) {
        int ox = 4 * (i8 & 1), oy = 2 * (i8 & 2);
        uint32_t e[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t o = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; hlf++) {
                int q = (k >> 1) * 2 + hlf;
                uint32_t t = chroma_row4(F, lane, L->mb_x * 8 + ox + 2 * hlf, L->mb_y * 8 + oy + k, mv4x[q], mv4y[q]);
                o |= (t & 0xFFFFu) << (16 * hlf);
            }
            r[k] = o;
            e[k] = lds4(L->fenc + 256 + (oy + k) * 16 + lane * 8 + ox);
        }
        if (satd) v = hadamard4x4_abs(e, r) >> 1;
        else { uint32_t s = 0; for (int k = 0; k < 4; k++) s = __builtin_amdgcn_sad_u8(e[k], r[k], s); v = (int)s; }
    }
    v = group_sum(v, 2);
    return __builtin_amdgcn_readlane(v, 0);
}

__device__ __forceinline__ void prim_load_fenc(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { int row = lane >> 2, c4 = lane & 3;
      sts4(L->fenc + row * 16 + c4 * 4, *(const uint32_t *)(F.fenc[0] + (size_t)(L->mb_y * 16 + row) * F.w + L->mb_x * 16 + c4 * 4)); }
    if (lane < 32) {
        int plane = lane >> 4, row = (lane & 15) >> 1, c4 = lane & 1;
        sts4(L->fenc + 256 + row * 16 + plane * 8 + c4 * 4,
             *(const uint32_t *)((plane ? F.fenc[2] : F.fenc[1]) + (size_t)(L->mb_y * 8 + row) * (F.w >> 1) + L->mb_x * 8 + c4 * 4));
    }
    PCAMV_WAVE_SYNC();
}

/* inter prediction of the whole MB from the per-4x4 MVs in L->cmv (x264_mb_mc, common/macroblock.c:483-508,626-690) */
/* one row of 4 luma pixels at quarter-pel MV (get_ref, mc.c:194-243) */
__device__ __forceinline__ uint32_t luma_row4(const FrameDev &F, int gx, int gy, int mvx, int mvy)
{
    const gp8 lb = (gp8)F.luma_base;
    const uint32_t stride = PCAMV_LROW, psz = (uint32_t)F.plane_size;
    const uint32_t o = (uint32_t)(gy + PCAMV_PAD + (mvy >> 2)) * stride + lsw_x((uint32_t)(gx + PCAMV_PAD + (mvx >> 2)), (uint32_t)F.lskip);
    const int dx = mvx & 3, dy = mvy & 3;
    uint32_t r = gld4(lb, o + (dx != 0 ? psz : 0u) + (dy == 2 ? 2u * psz : 0u) + (dy == 3 ? stride : 0u));
    if ((dx | dy) & 1) r = avg4(r, gld4(lb, o + (dy ? (dx == 2 ? 3u * psz : 2u * psz) : 0u) + (dx == 3)));
    return r;
}
/* two horizontally adjacent chroma pixels of mc_chroma at chroma position (cx,cy), in bits 0..15 */
__device__ __forceinline__ uint32_t chroma_px2(const FrameDev &F, int plane, int cx, int cy, int mvx, int mvy)
{
    const gp8 cb = (gp8)F.chroma_base[0];
    const uint32_t cstride = (uint32_t)F.cstride;
    const uint32_t o = (plane ? (uint32_t)F.cplane_size : 0u) + mul24u((uint32_t)(cy + PCAMV_CPAD + (mvy >> 3)), cstride) + (uint32_t)(cx + PCAMV_CPAD + (mvx >> 3));
    const int dx = mvx & 7, dy = mvy & 7;
    const uint32_t W = mul24u(8 - dx, 8 - dy) | mul24u(dx, 8 - dy) << 8 | mul24u(8 - dx, dy) << 16 | mul24u(dx, dy) << 24;
    const uint32_t a = gld4(cb, o), b = gld4(cb + cstride, o);
    const uint32_t p0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(b, a, 0x05040100u), W, 32u, false) >> 6;
    const uint32_t p1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(b, a, 0x06050201u), W, 32u, false) >> 6;
    return p0 | p1 << 8;
}
__device__ __forceinline__ void prim_predict_mb(const FrameDev &F, MBLocal *L, int win)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { int row = lane >> 2, c4 = lane & 3, i8 = SCAN8_0 + c4 + 8 * (row >> 2);
      int mvx = clip3i(L->cmv[i8][0], L->mv_min[0], L->mv_max[0]), mvy = clip3i(L->cmv[i8][1], L->mv_min[1], L->mv_max[1]);
      uint32_t v;
      if (win) {
          const int dx = mvx & 3, dy = mvy & 3;
          const int wb = (L->mb_y * 16 + row + PCAMV_PAD + (mvy >> 2) - L->win_y0) * WIN_LW + (L->mb_x * 16 + 4 * c4 + PCAMV_PAD + (mvx >> 2) - L->win_x0);
          v = wld4(L, wb + ((dx != 0) + 2 * (dy == 2)) * WIN_LP + (dy == 3 ? WIN_LW : 0));
          if ((dx | dy) & 1) v = avg4(v, wld4(L, wb + (dy ? (2 + (dx == 2)) * WIN_LP : 0) + (dx == 3)));
      } else v = luma_row4(F, L->mb_x * 16 + 4 * c4, L->mb_y * 16 + row, mvx, mvy);
      sts4(L->pred + row * 16 + 4 * c4, v); }
    {   /* chroma: every lane two pixels (one 2x2 chroma block row carries one luma 4x4's MV) */
        int plane = lane >> 5, row = (lane & 31) >> 2, c2 = lane & 3;
        int i8 = SCAN8_0 + c2 + 8 * (row >> 1);
        int mvx = clip3i(L->cmv[i8][0], L->mv_min[0], L->mv_max[0]), mvy = clip3i(L->cmv[i8][1], L->mv_min[1], L->mv_max[1]);
        uint32_t t;
        if (win) {
            const int b = 4 * WIN_LP + plane * WIN_CP + (L->mb_y * 8 + row + PCAMV_CPAD + (mvy >> 3) - L->win_cy0) * WIN_CW + (L->mb_x * 8 + 2 * c2 + PCAMV_CPAD + (mvx >> 3) - L->win_cx0);
            const int dx = mvx & 7, dy = mvy & 7;
            const uint32_t W = mul24u(8 - dx, 8 - dy) | mul24u(dx, 8 - dy) << 8 | mul24u(8 - dx, dy) << 16 | mul24u(dx, dy) << 24;
            const uint32_t a = wld4(L, b), bb = wld4(L, b + WIN_CW);
            t = (__builtin_amdgcn_udot4(__builtin_amdgcn_perm(bb, a, 0x05040100u), W, 32u, false) >> 6)
              | (__builtin_amdgcn_udot4(__builtin_amdgcn_perm(bb, a, 0x06050201u), W, 32u, false) >> 6) << 8;
        } else t = chroma_px2(F, plane, L->mb_x * 8 + 2 * c2, L->mb_y * 8 + row, mvx, mvy);
        *(uint16_t *)(L->pred + 256 + row * 16 + plane * 8 + 2 * c2) = (uint16_t)t;
    }
    PCAMV_WAVE_SYNC();
}
/* which: 0 luma only, 1 luma+chroma, 2 chroma only; (mvx,mvy) already clipped */
__device__ __forceinline__ void prim_predict_16x16(const FrameDev &F, MBLocal *L, int mvx, int mvy, int which)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    if (which != 2) { int row = lane >> 2, c4 = lane & 3; sts4(L->pred + row * 16 + 4 * c4, luma_row4(F, L->mb_x * 16 + 4 * c4, L->mb_y * 16 + row, mvx, mvy)); }
    if (which != 0) {
        int plane = lane >> 5, row = (lane & 31) >> 2, c2 = lane & 3;
        uint32_t t = chroma_px2(F, plane, L->mb_x * 8 + 2 * c2, L->mb_y * 8 + row, mvx, mvy);
        *(uint16_t *)(L->pred + 256 + row * 16 + plane * 8 + 2 * c2) = (uint16_t)t;
    }
    PCAMV_WAVE_SYNC();
}

/* forward 4x4 transform + quantisation + scan score + dequantisation of the lane's 4x4 block at
 * (px,py) of the fenc/pred buffers (dct.c:122-170 sub4x4_dct, quant.c:33-109, 203-239).  d[] returns the
 * DEquantised levels (all zero when nothing survives), *rawdc the unquantised DC (chroma: its DC goes
 * through the 2x2 transform instead and d[0] is cleared before quantisation). */
__device__ __forceinline__ void quant_score_dequant(const FrameDev &F, bool is_l, int16_t d[16], int *nz_out, int *score_out, int *rawdc, int16_t *lv_out = nullptr)
{
    *rawdc = d[0];
    if (!is_l) d[0] = 0;
    const int qp = is_l ? F.qp : F.chroma_qp;
    /* the three position classes of the flat quantiser, picked once (no indexing of F inside the loops) */
    const int mf0 = is_l ? F.q_mf[0][0] : F.q_mf[1][0], mf1 = is_l ? F.q_mf[0][1] : F.q_mf[1][1], mf2 = is_l ? F.q_mf[0][2] : F.q_mf[1][2];
    const int bs0 = is_l ? F.q_bias[0][0] : F.q_bias[1][0], bs1 = is_l ? F.q_bias[0][1] : F.q_bias[1][1], bs2 = is_l ? F.q_bias[0][2] : F.q_bias[1][2];
    const int dq0 = is_l ? F.dq_mf[0] : F.dq_mf_c[0], dq1 = is_l ? F.dq_mf[1] : F.dq_mf_c[1], dq2 = is_l ? F.dq_mf[2] : F.dq_mf_c[2];
    /* quantise (quant.c:33-48: c > 0 ? (bias + c) * mf >> 16 : -((bias - c) * mf >> 16)) as ONE signed multiply-add and an arithmetic
     * shift per coefficient: with bm = bias * mf (< 2^16, so c = 0 stays 0) the positive arm is (c * mf + bm) >> 16, and the negative one,
     * -floor(x / 2^16) with x = |c| * mf + bm, is (c * mf - bm + 65535) >> 16; |c| < 2^14, mf < 2^16.  The non-zero mask in zigzag order
     * and the "some |level| > 1" flag come from the levels afterwards (maximum / minimum of all sixteen). */
    const int bm0 = (int)mul24u((uint32_t)bs0, (uint32_t)mf0), bm1 = (int)mul24u((uint32_t)bs1, (uint32_t)mf1), bm2 = (int)mul24u((uint32_t)bs2, (uint32_t)mf2);
    unsigned zm = 0; int mx = 0, mn = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        constexpr int zzinv[16] = {0, 2, 3, 9, 1, 4, 8, 10, 5, 7, 11, 14, 6, 12, 13, 15};   /* raster index -> scan position (inverse of the zigzag) */
        const int cls = (i & 1) + ((i >> 2) & 1), mf = cls == 0 ? mf0 : cls == 1 ? mf1 : mf2, bm = cls == 0 ? bm0 : cls == 1 ? bm1 : bm2;
        const int c = d[i];
        const int q = mad24s(c, mf, c < 0 ? 65535 - bm : bm) >> 16;
        d[i] = (int16_t)q;
        zm |= q != 0 ? 1u << zzinv[i] : 0u;
        mx = imax(mx, q); mn = imin(mn, q);
    }
    const int big = mx > 1 || mn < -1;
    const int nz = zm != 0;
    if (lv_out && nz) {       /* the quantised levels in scan order (h->dct.luma4x4), for the entropy coder's size walk */
        constexpr int zz[16] = {0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15};
#pragma unroll
        for (int k = 0; k < 16; k += 2) *(uint32_t *)(lv_out + k) = (uint32_t)(uint16_t)d[zz[k]] | (uint32_t)(uint16_t)d[zz[k + 1]] << 16;
    }
    /* x264_decimate_score (quant.c:203-239) on the zigzag scan (16 coefficients for luma, the 15 AC ones for chroma): 9 as soon as a
     * level exceeds 1, else the table {3, 2, 2, 1, 1, 1, 0, ..}[run of zeros below it] summed over the non-zero levels.  Without a
     * loop: the table is (run < 1) + (run < 3) + (run < 6), and "the run below position p is shorter than k" is "a level sits in one of
     * the k positions below p" -- with a virtual level just below the scan's first position (bit 0 of z) that is bit p of
     * z << 1 | .. | z << k, so each of the three terms is one population count. */
    int score = 0;
    if (nz) {
        const unsigned zr = is_l ? zm << 1 : zm, z = zr | 1u;           /* (chroma: position 0, the DC, is never set -- it is the virtual level) */
        const unsigned s1 = z << 1, s3 = s1 | s1 << 1 | z << 3, s6 = s3 | s3 << 3;
        score = big ? 9 : __builtin_popcount(zr & s1) + __builtin_popcount(zr & s3) + __builtin_popcount(zr & s6);
        const int qbits = qp / 6 - 4;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int cls = (i & 1) + ((i >> 2) & 1), dqv = cls == 0 ? dq0 : cls == 1 ? dq1 : dq2;
            d[i] = qbits >= 0 ? (int16_t)(mul24s(d[i], dqv) << qbits) : (int16_t)((mul24s(d[i], dqv) + (1 << (-qbits - 1))) >> (-qbits));
        }
    }
    *nz_out = nz; *score_out = score;
}
/* forward transform of the lane's 4x4 block at (px,py): fenc minus the prediction buffer pred (dct.c:122-170) */
__device__ __forceinline__ void residual_block_at(const FrameDev &F, MBLocal *L, const uint8_t *pred, int px, int py, bool is_l, int16_t d[16], int *nz_out, int *score_out, int *rawdc, int16_t *lv_out = nullptr)
{
    int t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t e = lds4(L->fenc + (py + y) * 16 + px), p = lds4(pred + (py + y) * 16 + px);
        int d0 = (int)(e & 255) - (int)(p & 255), d1 = (int)((e >> 8) & 255) - (int)((p >> 8) & 255);
        int d2 = (int)((e >> 16) & 255) - (int)((p >> 16) & 255), d3 = (int)(e >> 24) - (int)(p >> 24);
        int s03 = d0 + d3, s12 = d1 + d2, d03 = d0 - d3, d12 = d1 - d2;
        t[0][y] = s03 + s12; t[1][y] = 2 * d03 + d12; t[2][y] = s03 - s12; t[3][y] = d03 - 2 * d12;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int s03 = t[i][0] + t[i][3], s12 = t[i][1] + t[i][2], d03 = t[i][0] - t[i][3], d12 = t[i][1] - t[i][2];
        d[i * 4 + 0] = (int16_t)(s03 + s12); d[i * 4 + 1] = (int16_t)(2 * d03 + d12); d[i * 4 + 2] = (int16_t)(s03 - s12); d[i * 4 + 3] = (int16_t)(d03 - 2 * d12);
    }
    quant_score_dequant(F, is_l, d, nz_out, score_out, rawdc, lv_out);
}
__device__ __forceinline__ void residual_block(const FrameDev &F, MBLocal *L, int px, int py, bool is_l, int16_t d[16], int *nz_out, int *score_out, int *rawdc, int16_t *lv_out = nullptr)
{
    residual_block_at(F, L, L->pred, px, py, is_l, d, nz_out, score_out, rawdc, lv_out);
}
/* one 4x4 block per lane into LDS (lanes 0..15 luma blocks in x264 block order, 16..19 U, 20..23 V):
 * the form the P_SKIP probe's wave-uniform checks read */
__device__ __forceinline__ void prim_residual(const FrameDev &F, MBLocal *L, int do_luma, int do_chroma)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    const bool is_l = lane < 16 && do_luma, is_c = lane >= 16 && lane < 24 && do_chroma;
    if (is_l || is_c) {
        int16_t d[16];
        int ch = (lane - 16) >> 2, ci = (lane - 16) & 3, nz, score, rawdc;
        int px = is_l ? 4 * blk_x_of(lane) : ch * 8 + (ci & 1) * 4;
        int py = is_l ? 4 * blk_y_of(lane) : 16 + (ci >> 1) * 4;
        residual_block(F, L, px, py, is_l, d, &nz, &score, &rawdc);
        if (is_c) L->red[lane] = rawdc;
#pragma unroll
        for (int i = 0; i < 16; i += 2) *(uint32_t *)&L->coef[lane][i] = (uint32_t)(uint16_t)d[i] | (uint32_t)(uint16_t)d[i + 1] << 16;
        L->blk_nz[lane] = nz; L->blk_score[lane] = score;
    }
    PCAMV_WAVE_SYNC();
    if (is_c && ((lane - 16) & 3) == 0) {
        /* dct2x2dc over the four raw DCs of this plane */
        int ch = (lane - 16) >> 2;
        int b0 = L->red[lane], b1 = L->red[lane + 1], b2 = L->red[lane + 2], b3 = L->red[lane + 3];
        int d0 = b0 + b1, d1 = b2 + b3, d2 = b0 - b1, d3 = b2 - b3;
        L->cdc[ch][0] = (int16_t)(d0 + d1); L->cdc[ch][1] = (int16_t)(d0 - d1); L->cdc[ch][2] = (int16_t)(d2 + d3); L->cdc[ch][3] = (int16_t)(d2 - d3);
    }
    PCAMV_WAVE_SYNC();
}

__device__ __forceinline__ void idct4x4_add(uint8_t *dst, const int16_t *c)   /* dst stride 16 */
{
    int16_t t[4][4], r[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int s02 = c[0 * 4 + i] + c[2 * 4 + i], d02 = c[0 * 4 + i] - c[2 * 4 + i];
        int s13 = c[1 * 4 + i] + (c[3 * 4 + i] >> 1), d13 = (c[1 * 4 + i] >> 1) - c[3 * 4 + i];
        t[i][0] = (int16_t)(s02 + s13); t[i][1] = (int16_t)(d02 + d13); t[i][2] = (int16_t)(d02 - d13); t[i][3] = (int16_t)(s02 - s13);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int s02 = t[0][i] + t[2][i], d02 = t[0][i] - t[2][i];
        int s13 = t[1][i] + (t[3][i] >> 1), d13 = (t[1][i] >> 1) - t[3][i];
        r[0][i] = (int16_t)((s02 + s13 + 32) >> 6); r[1][i] = (int16_t)((d02 + d13 + 32) >> 6);
        r[2][i] = (int16_t)((d02 - d13 + 32) >> 6); r[3][i] = (int16_t)((s02 - s13 + 32) >> 6);
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t p = lds4(dst + y * 16), o = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((p >> (8 * x)) & 255) + r[y][x], 0, 255) << (8 * x);
        sts4(dst + y * 16, o);
    }
}
__device__ __forceinline__ void prim_add_idct(const FrameDev &F, MBLocal *L, unsigned keep, int cm0, int cm1)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    if (lane < 16) {
        if (((keep >> lane) & 1) && L->blk_nz[lane]) {
            int16_t c[16];
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = L->coef[lane][i];
            idct4x4_add(L->pred + 4 * blk_y_of(lane) * 16 + 4 * blk_x_of(lane), c);
        }
    } else if (lane < 24) {
        int ch = (lane - 16) >> 2, ci = (lane - 16) & 3, mode = ch ? cm1 : cm0;
        uint8_t *dst = L->pred + 256 + (ci >> 1) * 4 * 16 + ch * 8 + (ci & 1) * 4;
        if (mode == 2) {
            int16_t c[16];
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = L->coef[lane][i];
            idct4x4_add(dst, c);
        } else if (mode == 1) {
            int v = ((int)(int16_t)L->cdc[ch][ci] + 32) >> 6;
#pragma unroll
            for (int y = 0; y < 4; y++) {
                uint32_t p = lds4(dst + y * 16), o = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((p >> (8 * x)) & 255) + v, 0, 255) << (8 * x);
                sts4(dst + y * 16, o);
            }
        }
    }
    (void)F;
    PCAMV_WAVE_SYNC();
}
/* 2x2 Hadamard over the four lanes of a quad: lane q gets v0 + s1 v1 + s2 v2 + s1 s2 v3 with
 * s1 = -1 for q in {2,3}, s2 = -1 for q in {1,3} (dct2x2dc / idct_dequant_2x2_dc, encoder/macroblock.c:71-85) */
__device__ __forceinline__ int quad_had2x2(int v, int q)
{
    int v0 = __builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, false), v1 = __builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, false);
    int v2 = __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, false), v3 = __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, false);
    int a = v0 + v1, b = v2 + v3, c = v0 - v1, e = v2 - v3;      /* d0, d1, d2, d3 of the reference */
    return q == 0 ? a + b : q == 1 ? a - b : q == 2 ? c + e : c - e;
}
/* Transform stage of x264_macroblock_encode for an inter macroblock (encoder/macroblock.c:277-372,
 * 696-753): residual transform + quantisation, luma 8x8 / macroblock decimation, chroma DC 2x2 and the
 * chroma decimation rule, dequantisation, inverse transform added to the prediction in L->pred.
 * One 4x4 block per lane (0..15 luma in x264 block order = quads of lanes are 8x8 blocks, 16..19 U,
 * 20..23 V), levels stay in registers, every decision is a quad / row DPP reduction:
 *   - the reference's saturating `if (dec8 < 6) dec8 += score` only ever compares against 4 and 6, so
 *     plain sums decide identically. */
/* lv: also leave the quantised levels in scan order (L->coef per block, L->cdc chroma DC in zigzag_scan_2x2_dc order) and the
 * per-block non-zero flags as the entropy coder sees them (L->nzc: zero where an 8x8 / the macroblock / a chroma plane was dropped) */
__device__ __forceinline__ void prim_mb_transform_v1(const FrameDev &F, MBLocal *L, int lv_ = 0)
{
    PCAMV_WAVE_SYNC();
    const int lv = rfl(lv_);
    const int lane = LANE();
    const bool is_l = lane < 16, is_c = lane >= 16 && lane < 24;
    const int ch = (lane - 16) >> 2, ci = (lane - 16) & 3;
    const int px = is_l ? 4 * blk_x_of(lane) : ch * 8 + (ci & 1) * 4;
    const int py = is_l ? 4 * blk_y_of(lane) : 16 + (ci >> 1) * 4;
    int16_t d[16];
    int nz = 0, score = 0, rawdc = 0;
    if (is_l || is_c) residual_block(F, L, px, py, is_l, d, &nz, &score, &rawdc, lv ? L->coef[lane < 24 ? lane : 0] : nullptr);
    /* luma: 8x8 sums over quads, macroblock sum over the row of 16 lanes */
    const int sc = (nz && F.b_dct_decimate) ? score : 0;
    int q8 = sc + dpp_qp1(sc); q8 += dpp_qp2(q8);
    int any8 = nz | dpp_qp1(nz); any8 |= dpp_qp2(any8);
    int row = q8 + dpp_hmir(q8); row += dpp_mir(row);            /* the four quad sums of the 16-lane row together */
    bool keep = F.b_dct_decimate ? (q8 >= 4 && row >= 6) : any8 != 0;
    /* chroma: per plane (quad) AC score, DC 2x2 transform, quantisation, dequantisation */
    const int cdc = quad_had2x2(rawdc, lane & 3);
    int dcq;
    { const int mf = F.q_mf[1][0] >> 1, bias = F.q_bias[1][0] << 1;
      dcq = cdc > 0 ? ((bias + cdc) * mf >> 16) : -((bias - cdc) * mf >> 16); }
    int nzdc = dcq != 0; nzdc |= dpp_qp1(nzdc); nzdc |= dpp_qp2(nzdc);
    int dmf = F.dq_mf_c[0], qbits = F.chroma_qp / 6 - 5;
    if (qbits > 0) { dmf <<= qbits; qbits = 0; }
    const int rdc = (int16_t)(quad_had2x2(dcq, lane & 3) * dmf >> -qbits);
    const int cmode = (q8 < 7 && F.b_dct_decimate) || !any8 ? (nzdc ? 1 : 0) : 2;     /* on chroma lanes q8 / any8 are the plane's */
    const unsigned long long keep_mask = __ballot(is_l && keep), ac_mask = __ballot(is_c && cmode == 2);
    uint8_t *dst = L->pred + py * 16 + px;
    if (is_l) {
        if (keep && nz) idct4x4_add(dst, d);
    } else if (is_c) {
        if (cmode == 2) { if (nzdc) d[0] = (int16_t)rdc; idct4x4_add(dst, d); }
        else if (cmode == 1) {
            const int v = (rdc + 32) >> 6;
#pragma unroll
            for (int y = 0; y < 4; y++) {
                uint32_t p = lds4(dst + y * 16), o = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((p >> (8 * x)) & 255) + v, 0, 255) << (8 * x);
                sts4(dst + y * 16, o);
            }
        }
    }
    L->nnz_mask = (int)(__ballot(is_l && keep && nz) & 0xffffu);
    L->cbp_luma = (int)((keep_mask & 1) | ((keep_mask >> 3) & 2) | ((keep_mask >> 6) & 4) | ((keep_mask >> 9) & 8));
    const unsigned long long dc_mask = __ballot(is_c && nzdc);
    L->cbp_chroma = ac_mask ? 2 : dc_mask ? 1 : 0;                 /* encoder/macroblock.c:364-372: DC-only chroma */
    if (lv) {
        if (is_l) L->nzc[scan8_of(lane)] = (uint8_t)(keep && nz);
        else if (is_c) {
            L->nzc[scan8_all_of(lane)] = (uint8_t)(cmode == 2 && nz);
            const int q = lane & 3;
            L->cdc[ch][q == 1 ? 2 : q == 2 ? 1 : q] = (int16_t)dcq;     /* zigzag_scan_2x2_dc: d[0][0], d[1][0], d[0][1], d[1][1] */
            if (q == 0) L->nzc[scan8_all_of(25 + ch)] = (uint8_t)(nzdc != 0);
        }
        if (lane == 0) L->nzc[scan8_all_of(24)] = 0;
    }
    PCAMV_WAVE_SYNC();
}
/* lane ^ 4 inside a row of 16: two row shifts, each written to the banks (quads) it is right for */
__device__ __forceinline__ int dpp_x4(int v)
{
    const int a = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);
    return __builtin_amdgcn_update_dpp(a, v, 0x114, 0xf, 0xa, false);
}
/* The same stage with TWO lanes per 4x4 block (round 3): a lone macroblock has 24 blocks, one per lane left 40 lanes idle through ~900
 * instructions.  Lane 2 b + h (luma, b in x264 block order: an 8x8 is eight lanes) and 32 + 2 cb + h (chroma, a plane is eight lanes)
 * holds rows 0 and 3 (h = 0) or 1 and 2 (h = 1) of its block: the horizontal transform and the first vertical butterfly (row A +- row B)
 * are in-lane, the second one exchanges with the partner lane (h = 0 ends with the coefficients of vertical frequency 0 and 1, h = 1 with
 * 2 and 3, all four horizontal frequencies each); quantisation, scan masks and dequantisation work on eight coefficients instead of
 * sixteen; the inverse transform runs the other way round (horizontal pass in-lane per vertical frequency, one exchange, and h = 0 ends
 * with rows 0 and 3, h = 1 with rows 1 and 2 -- the rows it loaded).  Same results as prim_mb_transform_v1, the arithmetic is the
 * reference's (dct.c:122-170, quant.c, macroblock.c) value for value. */
__device__ __forceinline__ void prim_mb_transform(const FrameDev &F, MBLocal *L, int lv_ = 0)
{
    PCAMV_WAVE_SYNC();
    const int lv = rfl(lv_);
    const int lane = LANE();
    const bool is_l = lane < 32, is_c = lane >= 32 && lane < 48;
    const int h = lane & 1, blk = is_l ? lane >> 1 : 16 + ((lane - 32) >> 1);          /* 0..15 luma, 16..19 U, 20..23 V (the index of L->coef / nzc) */
    const int cb = (lane - 32) >> 1, ch = cb >> 2, ci = cb & 3;
    const int px = is_l ? 4 * blk_x_of(blk) : ch * 8 + (ci & 1) * 4;
    const int py = is_l ? 4 * blk_y_of(blk) : 16 + (ci >> 1) * 4;
    const int ra = h ? 1 : 0, rb = h ? 2 : 3;                                      /* this lane's two rows */
    const int sgn = h ? -1 : 1;
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};           /* coefficients: c[j] = (horizontal frequency j, vertical frequency 2 h), c[4 + j] = (j, 2 h + 1) */
    int nz = 0, score = 0, rawdc = 0, big = 0;
    unsigned zm = 0;
    if (is_l || is_c) {
        int t[2][4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int y = k ? rb : ra;
            const uint32_t e = lds4(L->fenc + (py + y) * 16 + px), p = lds4(L->pred + (py + y) * 16 + px);
            const int d0 = (int)(e & 255) - (int)(p & 255), d1 = (int)((e >> 8) & 255) - (int)((p >> 8) & 255);
            const int d2 = (int)((e >> 16) & 255) - (int)((p >> 16) & 255), d3 = (int)(e >> 24) - (int)(p >> 24);
            const int s03 = d0 + d3, s12 = d1 + d2, d03 = d0 - d3, d12 = d1 - d2;
            t[k][0] = s03 + s12; t[k][1] = 2 * d03 + d12; t[k][2] = s03 - s12; t[k][3] = d03 - 2 * d12;
        }
        /* vertical: rows (0, 3) / (1, 2) in-lane, then with the partner: h = 0: s03 + s12, 2 d03 + d12; h = 1: s03 - s12, d03 - 2 d12 */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int s = t[0][j] + t[1][j], d = t[0][j] - t[1][j];
            c[j] = mad24s(s, sgn, dpp_qp1(s));
            c[4 + j] = mad24s(d, 2 * sgn, dpp_qp1(d));
        }
    }
    rawdc = h ? 0 : c[0];
    if (!is_l && !h) c[0] = 0;
    {
        /* quantiser and dequantiser of the lane's eight positions: class (j & 1) + (vertical frequency & 1) -- c[0..3]: 0 1 0 1, c[4..7]: 1 2 1 2 */
        const int mf0 = is_l ? F.q_mf[0][0] : F.q_mf[1][0], mf1 = is_l ? F.q_mf[0][1] : F.q_mf[1][1], mf2 = is_l ? F.q_mf[0][2] : F.q_mf[1][2];
        const int bs0 = is_l ? F.q_bias[0][0] : F.q_bias[1][0], bs1 = is_l ? F.q_bias[0][1] : F.q_bias[1][1], bs2 = is_l ? F.q_bias[0][2] : F.q_bias[1][2];
        const int dq0 = is_l ? F.dq_mf[0] : F.dq_mf_c[0], dq1 = is_l ? F.dq_mf[1] : F.dq_mf_c[1], dq2 = is_l ? F.dq_mf[2] : F.dq_mf_c[2];
        const int bm0 = (int)mul24u((uint32_t)bs0, (uint32_t)mf0), bm1 = (int)mul24u((uint32_t)bs1, (uint32_t)mf1), bm2 = (int)mul24u((uint32_t)bs2, (uint32_t)mf2);
        /* scan positions (inverse zigzag of raster index 4 j + vertical frequency): vf 0: 0 1 5 6, 1: 2 4 7 12, 2: 3 8 11 13, 3: 9 10 14 15 */
        int mx = 0, mn = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = k & 3, odd = k >> 2, cls = (j & 1) + odd;
            const int mf = cls == 0 ? mf0 : cls == 1 ? mf1 : mf2, bm = cls == 0 ? bm0 : cls == 1 ? bm1 : bm2;
            constexpr int pos0[8] = {0, 1, 5, 6, 2, 4, 7, 12}, pos1[8] = {3, 8, 11, 13, 9, 10, 14, 15};
            const int q = mad24s(c[k], mf, c[k] < 0 ? 65535 - bm : bm) >> 16;
            c[k] = q;
            zm |= q != 0 ? (h ? 1u << pos1[k] : 1u << pos0[k]) : 0u;
            mx = imax(mx, q); mn = imin(mn, q);
        }
        big = mx > 1 || mn < -1;
        zm |= (unsigned)dpp_qp1((int)zm);
        big |= dpp_qp1(big);
        nz = zm != 0;
        if (lv && nz && (is_l || is_c)) {          /* the quantised levels in scan order (h->dct.luma4x4), for the entropy coder's size walk */
            int16_t *lv_out = L->coef[blk];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                constexpr int pos0[8] = {0, 1, 5, 6, 2, 4, 7, 12}, pos1[8] = {3, 8, 11, 13, 9, 10, 14, 15};
                lv_out[h ? pos1[k] : pos0[k]] = (int16_t)c[k];
            }
        }
        if (nz) {
            const unsigned zr = is_l ? zm << 1 : zm, z = zr | 1u;
            const unsigned s1 = z << 1, s3 = s1 | s1 << 1 | z << 3, s6 = s3 | s3 << 3;
            score = big ? 9 : __builtin_popcount(zr & s1) + __builtin_popcount(zr & s3) + __builtin_popcount(zr & s6);
            const int qbits = (is_l ? F.qp : F.chroma_qp) / 6 - 4;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int cls = (k & 1) + (k >> 2), dqv = cls == 0 ? dq0 : cls == 1 ? dq1 : dq2;
                c[k] = qbits >= 0 ? (int16_t)(mul24s(c[k], dqv) << qbits) : (int16_t)((mul24s(c[k], dqv) + (1 << (-qbits - 1))) >> (-qbits));
            }
        }
    }
    /* luma: 8x8 sums over the eight lanes of an 8x8, macroblock sum over lanes 0..31; chroma: the plane's eight lanes */
    const int sc = (nz && F.b_dct_decimate && !h) ? score : 0;
    int q8 = sc + dpp_qp1(sc); q8 += dpp_qp2(q8); q8 += dpp_hmir(q8);
    int any8 = nz | dpp_qp2(nz); any8 |= dpp_hmir(any8);
    const int r16 = q8 + dpp_mir(q8);
    const int row = __builtin_amdgcn_readlane(r16, 0) + __builtin_amdgcn_readlane(r16, 16);
    const bool keep = F.b_dct_decimate ? (q8 >= 4 && row >= 6) : any8 != 0;
    /* chroma DC: 2x2 transform over the plane's four blocks (their h = 0 lanes, two apart): butterflies with lane ^ 2, then lane ^ 4; the
     * network leaves coefficient (k & 1) * 2 + (k >> 1) in the lane of block k -- zigzag_scan_2x2_dc's place k -- and, run again on the
     * quantised values, block k's reconstructed DC in the lane of block k */
    const int s2 = lane & 2 ? -1 : 1, s4 = lane & 4 ? -1 : 1;
    int cdc = mad24s(rawdc, s2, dpp_qp2(rawdc));
    cdc = mad24s(cdc, s4, dpp_x4(cdc));
    int dcq;
    { const int mf = F.q_mf[1][0] >> 1, bias = F.q_bias[1][0] << 1;
      dcq = cdc > 0 ? ((bias + cdc) * mf >> 16) : -((bias - cdc) * mf >> 16); }
    if (h || !is_c) dcq = 0;
    int nzdc = dcq != 0; nzdc |= dpp_qp1(nzdc); nzdc |= dpp_qp2(nzdc); nzdc |= dpp_hmir(nzdc);
    int dmf = F.dq_mf_c[0], qbits = F.chroma_qp / 6 - 5;
    if (qbits > 0) { dmf <<= qbits; qbits = 0; }
    int idc = mad24s(dcq, s2, dpp_qp2(dcq));
    idc = mad24s(idc, s4, dpp_x4(idc));
    int rdc = (int16_t)(idc * dmf >> -qbits);
    {   /* both lanes of the block (the exchange made by all lanes, THEN the choice: inside `h ? dpp : x` only the odd lanes would
         * execute it, and a DPP read of a lane that is switched off returns 0) */
        const int other = dpp_qp1(rdc);
        rdc = h ? other : rdc;
    }
    const int cmode = (q8 < 7 && F.b_dct_decimate) || !any8 ? (nzdc ? 1 : 0) : 2;
    const unsigned long long keep_mask = __ballot(is_l && keep), ac_mask = __ballot(is_c && cmode == 2);
    const bool inv = is_l ? keep && nz : is_c && cmode == 2;
    if (is_c && cmode == 2 && nzdc && !h) c[0] = (int16_t)rdc;
    if (inv) {
        /* inverse: per vertical frequency the horizontal pass (in-lane), then the vertical one with the partner */
        int u[2][4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c0 = c[4 * k], c1 = c[4 * k + 1], c2 = c[4 * k + 2], c3 = c[4 * k + 3];
            const int s02 = c0 + c2, d02 = c0 - c2, s13 = c1 + (c3 >> 1), d13 = (c1 >> 1) - c3;
            u[k][0] = (int16_t)(s02 + s13); u[k][1] = (int16_t)(d02 + d13); u[k][2] = (int16_t)(d02 - d13); u[k][3] = (int16_t)(s02 - s13);
        }
        uint32_t oa = 0, ob = 0;
        const uint8_t *pa = L->pred + (py + ra) * 16 + px, *pb = L->pred + (py + rb) * 16 + px;
        const uint32_t va = lds4(pa), vb = lds4(pb);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            /* h = 0 holds t[0][i], t[1][i] and makes s02, s13; h = 1 holds t[2][i], t[3][i] and makes d02, d13 */
            const int x = mad24s(u[0][i], sgn, dpp_qp1(u[0][i])) + 32;
            const int y = mad24s(u[1][i], sgn, dpp_qp1(u[1][i]) >> 1);
            const int r_a = (int16_t)((x + y) >> 6), r_b = (int16_t)((x - y) >> 6);      /* rows 0 / 3 (h = 0), 1 / 2 (h = 1) */
            oa |= (uint32_t)clip3i((int)((va >> (8 * i)) & 255) + r_a, 0, 255) << (8 * i);
            ob |= (uint32_t)clip3i((int)((vb >> (8 * i)) & 255) + r_b, 0, 255) << (8 * i);
        }
        sts4((uint8_t *)pa, oa); sts4((uint8_t *)pb, ob);
    } else if (is_c && cmode == 1) {
        const int v = (rdc + 32) >> 6;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint8_t *dst = L->pred + (py + (k ? rb : ra)) * 16 + px;
            uint32_t p = lds4(dst), o = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((p >> (8 * x)) & 255) + v, 0, 255) << (8 * x);
            sts4(dst, o);
        }
    }
    {
        /* one bit per block out of the even lanes' bits */
        unsigned m = (unsigned)__ballot(is_l && keep && nz && !h);
        m = (m | m >> 1) & 0x33333333u; m = (m | m >> 2) & 0x0f0f0f0fu; m = (m | m >> 4) & 0x00ff00ffu; m = (m | m >> 8) & 0xffffu;
        L->nnz_mask = (int)m;
    }
    L->cbp_luma = (int)((keep_mask & 1) | ((keep_mask >> 7) & 2) | ((keep_mask >> 14) & 4) | ((keep_mask >> 21) & 8));
    const unsigned long long dc_mask = __ballot(is_c && nzdc);
    L->cbp_chroma = ac_mask ? 2 : dc_mask ? 1 : 0;                 /* encoder/macroblock.c:364-372: DC-only chroma */
    if (lv && !h) {
        if (is_l) L->nzc[scan8_of(blk)] = (uint8_t)(keep && nz);
        else if (is_c) {
            L->nzc[scan8_all_of(blk)] = (uint8_t)(cmode == 2 && nz);
            L->cdc[ch][ci] = (int16_t)dcq;                              /* (the network's place = zigzag_scan_2x2_dc's) */
            if (ci == 0) L->nzc[scan8_all_of(25 + ch)] = (uint8_t)(nzdc != 0);
        }
        if (lane == 0) L->nzc[scan8_all_of(24)] = 0;
    }
    PCAMV_WAVE_SYNC();
}
/* ---- four re-encodes of a 16x16 macroblock at once (RCA step, reference pixels in the LDS window) ---- */
/* 16x16 prediction with one MV out of the window into pred4[j] */
__device__ __forceinline__ void prim_predict_win16(const FrameDev &F, MBLocal *L, int j, int mvx, int mvy)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    uint8_t *dst = L->pred4[j];
    { const int row = lane >> 2, c4 = lane & 3, dx = mvx & 3, dy = mvy & 3;
      const int wb = (L->mb_y * 16 + row + PCAMV_PAD + (mvy >> 2) - L->win_y0) * WIN_LW + (L->mb_x * 16 + 4 * c4 + PCAMV_PAD + (mvx >> 2) - L->win_x0);
      uint32_t v = wld4(L, wb + ((dx != 0) + 2 * (dy == 2)) * WIN_LP + (dy == 3 ? WIN_LW : 0));
      if ((dx | dy) & 1) v = avg4(v, wld4(L, wb + (dy ? (2 + (dx == 2)) * WIN_LP : 0) + (dx == 3)));
      sts4(dst + row * 16 + 4 * c4, v); }
    { const int plane = lane >> 5, row = (lane & 31) >> 2, c2 = lane & 3, dx = mvx & 7, dy = mvy & 7;
      const int b = 4 * WIN_LP + plane * WIN_CP + (L->mb_y * 8 + row + PCAMV_CPAD + (mvy >> 3) - L->win_cy0) * WIN_CW + (L->mb_x * 8 + 2 * c2 + PCAMV_CPAD + (mvx >> 3) - L->win_cx0);
      const uint32_t W = mul24u(8 - dx, 8 - dy) | mul24u(dx, 8 - dy) << 8 | mul24u(8 - dx, dy) << 16 | mul24u(dx, dy) << 24;
      const uint32_t a = wld4(L, b), bb = wld4(L, b + WIN_CW);
      const uint32_t t = (__builtin_amdgcn_udot4(__builtin_amdgcn_perm(bb, a, 0x05040100u), W, 32u, false) >> 6)
                       | (__builtin_amdgcn_udot4(__builtin_amdgcn_perm(bb, a, 0x06050201u), W, 32u, false) >> 6) << 8;
      *(uint16_t *)(dst + 256 + row * 16 + plane * 8 + 2 * c2) = (uint16_t)t; }
    (void)F;
    PCAMV_WAVE_SYNC();
}
/* transform stage of the four predictions in pred4 (same rules as prim_mb_transform): pass A, every lane
 * one luma block (macroblock j = lane / 16: a DPP row each); pass B, lanes 0..31 one chroma block each
 * (macroblock j = lane / 8, plane = quad) */
__device__ __forceinline__ void prim_mb_transform4(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    {
        const int j = lane >> 4, blk = lane & 15, px = 4 * blk_x_of(blk), py = 4 * blk_y_of(blk);
        uint8_t *pr = L->pred4[j];
        int16_t d[16]; int nz, score, rawdc;
        residual_block_at(F, L, pr, px, py, true, d, &nz, &score, &rawdc);
        const int sc = (nz && F.b_dct_decimate) ? score : 0;
        int q8 = sc + dpp_qp1(sc); q8 += dpp_qp2(q8);
        int any8 = nz | dpp_qp1(nz); any8 |= dpp_qp2(any8);
        int row = q8 + dpp_hmir(q8); row += dpp_mir(row);
        const bool keep = F.b_dct_decimate ? (q8 >= 4 && row >= 6) : any8 != 0;
        if (keep && nz) idct4x4_add(pr + py * 16 + px, d);
    }
    if (lane < 32) {
        const int j = lane >> 3, ch = (lane >> 2) & 1, ci = lane & 3, px = ch * 8 + (ci & 1) * 4, py = 16 + (ci >> 1) * 4;
        uint8_t *pr = L->pred4[j];
        int16_t d[16]; int nz, score, rawdc;
        residual_block_at(F, L, pr, px, py, false, d, &nz, &score, &rawdc);
        const int sc = (nz && F.b_dct_decimate) ? score : 0;
        int q8 = sc + dpp_qp1(sc); q8 += dpp_qp2(q8);
        int any8 = nz | dpp_qp1(nz); any8 |= dpp_qp2(any8);
        const int cdc = quad_had2x2(rawdc, ci);
        int dcq;
        { const int mf = F.q_mf[1][0] >> 1, bias = F.q_bias[1][0] << 1;
          dcq = cdc > 0 ? ((bias + cdc) * mf >> 16) : -((bias - cdc) * mf >> 16); }
        int nzdc = dcq != 0; nzdc |= dpp_qp1(nzdc); nzdc |= dpp_qp2(nzdc);
        int dmf = F.dq_mf_c[0], qbits = F.chroma_qp / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        const int rdc = (int16_t)(quad_had2x2(dcq, ci) * dmf >> -qbits);
        const int cmode = (q8 < 7 && F.b_dct_decimate) || !any8 ? (nzdc ? 1 : 0) : 2;
        uint8_t *dst = pr + py * 16 + px;
        if (cmode == 2) { if (nzdc) d[0] = (int16_t)rdc; idct4x4_add(dst, d); }
        else if (cmode == 1) {
            const int v = (rdc + 32) >> 6;
#pragma unroll
            for (int y = 0; y < 4; y++) {
                uint32_t p = lds4(dst + y * 16), o = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) o |= (uint32_t)clip3i((int)((p >> (8 * x)) & 255) + v, 0, 255) << (8 * x);
                sts4(dst + y * 16, o);
            }
        }
    }
    PCAMV_WAVE_SYNC();
}

__device__ __forceinline__ int prim_chroma_ssd(const FrameDev &F, MBLocal *L, int ch)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    int v = 0;
    if (lane < 16) {
        int row = lane >> 1, c4 = lane & 1;
        uint32_t e = lds4(L->fenc + 256 + row * 16 + ch * 8 + c4 * 4), p = lds4(L->pred + 256 + row * 16 + ch * 8 + c4 * 4);
#pragma unroll
        for (int x = 0; x < 4; x++) { int dd = (int)((e >> (8 * x)) & 255) - (int)((p >> (8 * x)) & 255); v += dd * dd; }
    }
    v = group_sum(v, 16);
    (void)F;
    return __builtin_amdgcn_readlane(v, 0);
}
__device__ __forceinline__ void prim_copy_pred(MBLocal *L, uint8_t *dst)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    ((uint32_t *)dst)[lane] = ((const uint32_t *)L->pred)[lane];
    if (lane < 32) ((uint32_t *)dst)[64 + lane] = ((const uint32_t *)L->pred)[64 + lane];
    PCAMV_WAVE_SYNC();
}
/* wt: store write-through (agent scope), for pixels another wave reads in the same launch (pass 2 -> loop filter) */
__device__ __forceinline__ void prim_store_rec(const FrameDev &F, MBLocal *L, bool wt = false)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { int row = lane >> 2, c4 = lane & 3;
      uint32_t *d = (uint32_t *)(F.rec[0] + (size_t)(L->mb_y * 16 + row) * F.w + L->mb_x * 16 + c4 * 4);
      const uint32_t v = lds4(L->pred + row * 16 + c4 * 4);
      if (wt) NB_ST32(d, v); else *d = v; }
    if (lane < 32) {
        int plane = lane >> 4, row = (lane & 15) >> 1, c4 = lane & 1;
        uint32_t *d = (uint32_t *)((plane ? F.rec[2] : F.rec[1]) + (size_t)(L->mb_y * 8 + row) * (F.w >> 1) + L->mb_x * 8 + c4 * 4);
        const uint32_t v = lds4(L->pred + 256 + row * 16 + plane * 8 + c4 * 4);
        if (wt) NB_ST32(d, v); else *d = v;
    }
}
/* the macroblock's reconstruction back out of the frame (pass 2 of a macroblock whose motion the embedding left alone: its first-pass
 * reconstruction IS its final one) */
__device__ __forceinline__ void prim_load_rec(const FrameDev &F, MBLocal *L)
{
    PCAMV_WAVE_SYNC();
    const int lane = LANE();
    { int row = lane >> 2, c4 = lane & 3;
      sts4(L->pred + row * 16 + c4 * 4, *(const uint32_t *)(F.rec[0] + (size_t)(L->mb_y * 16 + row) * F.w + L->mb_x * 16 + c4 * 4)); }
    if (lane < 32) {
        int plane = lane >> 4, row = (lane & 15) >> 1, c4 = lane & 1;
        sts4(L->pred + 256 + row * 16 + plane * 8 + c4 * 4, *(const uint32_t *)((plane ? F.rec[2] : F.rec[1]) + (size_t)(L->mb_y * 8 + row) * (F.w >> 1) + L->mb_x * 8 + c4 * 4));
    }
    PCAMV_WAVE_SYNC();
}
__device__ __forceinline__ void prim_store_mvr(const FrameDev &F, MBLocal *L, int mvx, int mvy)
{
    if (LANE() == 0) NB_ST32(&F.mvr[2 * L->mb_xy], NB_PACK16(mvx, mvy));
}
PCAMV_DEV void predict_mv(MBLocal *L, int idx, int width, int mvp[2]);      /* pcamv_logic.h */
#include "pcamv_prims_rd_gpu.h"
#endif
