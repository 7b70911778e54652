/*
 * pcamv_kernels.hip.h -- the __global__ kernels (gfx950).
 *
 *   k_chroma_pad      reference chroma -> padded planes (x264_frame_expand_border, frame.c:246)
 *   k_hpel            reference luma -> 4 padded planes full/H/V/HV, LDS-tiled 6-tap filter
 *                     (hpel_filter mc.c:167-190 + both border expansions, frame.c:246-301, in
 *                     closed form: value(x,y) = filter(clamp(x,-4,W+3), clamp(y,-8,H+7)))
 *   k_search_diag     phase A for one anti-diagonal, one wavefront per macroblock
 *   k_rca             phase B, one wavefront per (macroblock, carrier slot)
 *   k_encode          phase C, one wavefront per macroblock
 *   k_embed_prepare   cover / cost assembly + MVC adjustment + message (encoder.c:1561-1840)
 *   k_stc_forward/backward  syndrome-trellis Viterbi (embed.h:309-548), 1024 states = 1024 lanes
 */
#ifndef PCAMV_KERNELS_HIP_H
#define PCAMV_KERNELS_HIP_H
#include "pcamv_common.h"
#include "pcamv_prims_gpu.h"
#include "pcamv_mbkernels.h"
#include "stc_mats.h"

/* The kernels every instance of the library shares (plane production, the second pass, the embedding stage, the common analysis
 * kernel, the probes) are compiled by ONE translation unit, pcamv_gpu.hip; the units of the --me tesa and RD instances take only the
 * templates and types from this header (round 2 compiled -- and shipped -- every kernel four times). */
#if !defined(PCAMV_TESA_TU) && !defined(PCAMV_RD_TU)
#define PCAMV_MAIN_TU 1
#endif

/* ------------------------------------------------------------------ plane production */
/* Every kernel is batched over independent closed GOPs: blockIdx.z (plane kernels) or blockIdx.y
 * (macroblock kernels) selects the GOP's FrameDev in a device array.  One launch then carries
 * the same dependency step of all GOPs, which is what fills the 256 CUs (a single 1080p frame
 * exposes at most 60 independent macroblocks at a time). */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(256) k_chroma_pad(const FrameDev *__restrict__ Fs)
{
    const FrameDev &F = Fs[blockIdx.z >> 1];
    const int pl = blockIdx.z & 1;
    const uint8_t *__restrict__ src = F.raw[1 + pl];
    uint8_t *__restrict__ dst = F.chroma_base[pl];
    const int w = F.w >> 1, h = F.h >> 1, cstride = F.cstride, clines = F.clines;
    /* 4 pixels per thread; pad and width are multiples of 4, so a group is inside the picture or one replicated pixel */
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
    if (x >= cstride || y >= clines) return;
    const uint8_t *rowp = src + (size_t)clip3i(y - PCAMV_CPAD, 0, h - 1) * w;
    const int gx = x - PCAMV_CPAD;
    uint32_t v;
    if (gx < 0 || gx >= w) v = rowp[gx < 0 ? 0 : w - 1] * 0x01010101u;
    else if (((uintptr_t)src & 3) == 0) v = *(const uint32_t *)(rowp + gx);
    else v = rowp[gx] | rowp[gx + 1] << 8 | rowp[gx + 2] << 16 | (uint32_t)rowp[gx + 3] << 24;
    *(uint32_t *)(dst + (size_t)y * cstride + x) = v;
}
#endif

/* (clamp_u8: pcamv_prims_gpu.h) */
/* The four luma planes full / H / V / HV of the reference frame, padded (x264_frame_filter + expand_border,
 * common/mc.c:455-507, frame.c:246-300; the filtered planes are defined 4 columns / 8 rows beyond the picture and
 * replicated from there).  One thread = 4 horizontally adjacent output pixels, walking HP_ROWS rows down: it keeps
 * the 6 source rows x 12 source columns its filters need in registers (three dwords a row, one new row per output
 * row), so a source byte is fetched once per thread and never goes through LDS; every store is a full dword and a
 * wave's stores are contiguous.  The output pixel groups are aligned with the picture (pad and width are multiples
 * of 4), so a group is either inside the filtered domain or entirely replicated from its edge pixel. */
#define HP_ROWS 16
#define HP_THREADS 128
__device__ __forceinline__ void hpel_load_row(const uint8_t *__restrict__ rowp, int eg0, int W, bool fast, uint32_t d[3])
{
    if (fast) {
        const uint32_t *q = (const uint32_t *)(rowp + eg0 - 4);
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (uint32_t)rowp[clip3i(eg0 - 4 + 4 * i + b, 0, W - 1)] << (8 * b);
            d[i] = v;
        }
    }
}
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(HP_THREADS) k_hpel(const FrameDev *__restrict__ Fs)
{
    const FrameDev &F = Fs[blockIdx.z];
    const uint8_t *__restrict__ src = F.raw[0];
    uint8_t *__restrict__ planes = F.luma_base;
    const int W = F.w, H = F.h, stride = F.stride, lines = F.lines;
    const int x0 = 4 * (blockIdx.x * HP_THREADS + threadIdx.x), yb = blockIdx.y * HP_ROWS;
    if (x0 >= stride) return;
    /* first picture column of the group whose values this group shows, and which of its bytes when replicated */
    const int gx = x0 - PCAMV_PAD, eg0 = clip3i(gx, -4, W);
    const int rep = gx < -4 ? 0 : gx > W ? 3 : -1;
    const bool fast = eg0 >= 4 && eg0 + 8 <= W && ((uintptr_t)src & 3) == 0;
    const size_t psz = (size_t)F.plane_size;
    const unsigned strip = PCAMV_LSTRIP_OF(x0);
    const size_t strip_o = (size_t)x0 + (size_t)strip * (size_t)F.lskip;
    const bool dup = strip > 0 && (unsigned)x0 == strip * PCAMV_LSW;
    uint32_t w[6][3];
    uint32_t of = 0, oh = 0, ov = 0, oc = 0;
    int ey_prev = 0;
    for (int y = yb; y < yb + HP_ROWS && y < lines; y++) {
        const int ey = clip3i(y - PCAMV_PAD, -8, H + 7);
        if (y == yb || ey != ey_prev) {
            if (y == yb) {
#pragma unroll
                for (int k = 0; k < 5; k++) hpel_load_row(src + (size_t)clip3i(ey - 2 + k, 0, H - 1) * W, eg0, W, fast, w[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) { w[k][0] = w[k + 1][0]; w[k][1] = w[k + 1][1]; w[k][2] = w[k + 1][2]; }
            }
            hpel_load_row(src + (size_t)clip3i(ey + 3, 0, H - 1) * W, eg0, W, fast, w[5]);
            ey_prev = ey;
            /* window positions 2..10 = picture columns eg0-2 .. eg0+6: unrounded vertical 6-tap of each, and row 2 itself */
            int v[9], b2[9];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int q = (j + 2) >> 2, sh = 8 * ((j + 2) & 3);
#define HPB(k) ((int)(w[k][q] >> sh & 255))
                v[j] = HPB(0) + HPB(5) - 5 * (HPB(1) + HPB(4)) + 20 * (HPB(2) + HPB(3));
                b2[j] = HPB(2);
#undef HPB
            }
            of = w[2][1]; oh = ov = oc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int th = b2[k] + b2[k + 5] - 5 * (b2[k + 1] + b2[k + 4]) + 20 * (b2[k + 2] + b2[k + 3]);
                const int tc = v[k] + v[k + 5] - 5 * (v[k + 1] + v[k + 4]) + 20 * (v[k + 2] + v[k + 3]);
                oh |= clamp_u8((th + 16) >> 5) << (8 * k);
                ov |= clamp_u8((v[k + 2] + 16) >> 5) << (8 * k);
                oc |= clamp_u8((tc + 512) >> 10) << (8 * k);
            }
            if (rep >= 0) {
                const int sh = 8 * rep;
                of = (of >> sh & 255) * 0x01010101u; oh = (oh >> sh & 255) * 0x01010101u;
                ov = (ov >> sh & 255) * 0x01010101u; oc = (oc >> sh & 255) * 0x01010101u;
            }
        }
        /* strip layout (pcamv_common.h): the group's place in its own strip and, for a strip's first group, the repeat at the end
         * of the strip before it */
        const size_t o = (size_t)y * PCAMV_LROW + strip_o;
        *(uint32_t *)(planes + o) = of;
        if (F.luma_raster) *(uint32_t *)(F.luma_raster + (size_t)y * stride + x0) = of;
        *(uint32_t *)(planes + psz + o) = oh;
        *(uint32_t *)(planes + 2 * psz + o) = ov;
        *(uint32_t *)(planes + 3 * psz + o) = oc;
        if (dup) {
            const size_t o2 = o - (size_t)F.lskip;
            *(uint32_t *)(planes + o2) = of;
            *(uint32_t *)(planes + psz + o2) = oh;
            *(uint32_t *)(planes + 2 * psz + o2) = ov;
            *(uint32_t *)(planes + 3 * psz + o2) = oc;
        }
    }
}
#endif

/* ------------------------------------------------------------------ analysis phases */
template <int TESA>
__global__ void __launch_bounds__(64) k_search_diag(const FrameDev *__restrict__ Fs, int d)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    const FrameDev F = Fs[blockIdx.y];
    /* MBs of the anti-diagonal x + 2y = d */
    int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    int y = y_lo + (int)blockIdx.x, x = d - 2 * y;
    if (y >= F.mb_h || x < 0 || x >= F.mb_w) return;
    mbk_search<TESA>(F, &L, &A, x, y);
}
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_rca(const FrameDev *__restrict__ Fs, int slots_per_mb)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    const FrameDev F = Fs[blockIdx.y];
    if (!F.embed) return;
    int xy = blockIdx.x / slots_per_mb, k = blockIdx.x - xy * slots_per_mb;
    if (xy >= F.n_mb) return;
    mbk_rca(F, &L, &A, xy, k);
}
#endif
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_encode(const FrameDev *__restrict__ Fs)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    const FrameDev F = Fs[blockIdx.y];
    if ((int)blockIdx.x >= F.n_mb) return;
    mbk_encode(F, &L, &A, blockIdx.x);
}
#endif

/* ------------------------------------------------------------------ pass 2 + loop filter */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_pass2_diag(const FrameDev *__restrict__ Fs, int d)
{
    __shared__ MBLocal L;
    const FrameDev F = Fs[blockIdx.y];
    int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    int y = y_lo + (int)blockIdx.x, x = d - 2 * y;
    if (y >= F.mb_h || x < 0 || x >= F.mb_w) return;
    mbk_pass2(F, &L, x, y);
}
#endif

/* H.264 Tables 8-16 / 8-17: alpha(indexA), beta(indexB), tc0(indexA, bS = 1..3) */
__device__ static const uint8_t dbk_alpha_dev[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
                                                     32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__device__ static const uint8_t dbk_beta_dev[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
                                                    9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__device__ static const int8_t dbk_tc0_dev[52][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1}, {0, 1, 1}, {1, 1, 1},
    {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3}, {2, 2, 3}, {2, 2, 4}, {2, 3, 4},
    {2, 3, 4}, {3, 3, 5}, {3, 4, 6}, {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10}, {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16},
    {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

/* Loop filter of one macroblock (x264_frame_deblock_row, common/frame.c:627-798, inter macroblocks, 4x4
 * transform, one QP): the macroblock and the 4 pixels left of / above it are staged in LDS, the 32
 * boundary strengths are computed one per lane, then the four vertical and the four horizontal edges are
 * filtered in order (one line per lane: 16 luma, 8 + 8 chroma on even edges) and the touched pixels go back.
 * Needs (x-1,y), (x,y-1) and (x+1,y-1) filtered: same anti-diagonal order as the search. */
struct DeblockLDS { uint8_t sy[20][24]; uint8_t sc[2][12][16]; uint8_t sbs[2][4][4]; };
/* Lo: the MBLocal pass 2 of this macroblock has just run in (same wave): its unfiltered reconstruction (pred), type, final
 * motion and non-zero flags are taken from there; nullptr: everything is in memory like the neighbours' */
__device__ __forceinline__ void mbk_deblock(const FrameDev &F, DeblockLDS *D, int mx, int my, const MBLocal *Lo = nullptr)
{
    const uint8_t *own = Lo ? Lo->pred : nullptr;
    uint8_t (*sy)[24] = D->sy;              /* rows / cols -4..15 of the macroblock at [r + 4][c + 4] */
    uint8_t (*sc)[12][16] = D->sc;          /* chroma rows / cols -4..7 */
    uint8_t (*sbs)[4][4] = D->sbs;
    const int lane = LANE(), xy = my * F.mb_w + mx, W = F.w, CW = F.w >> 1;
    const int gx = 16 * mx, gy = 16 * my, cgx = 8 * mx, cgy = 8 * my;
    /* stage: 20 rows x 5 dwords of luma, 2 x 12 rows x 3 dwords of chroma (nothing outside the picture) */
    for (int i = lane; i < 100; i += 64) {
        const int r = i / 5 - 4, c = (i % 5) * 4 - 4;
        if (own && r >= 0 && c >= 0) *(uint32_t *)&sy[r + 4][c + 4] = lds4(own + r * 16 + c);
        else if (gy + r >= 0 && gx + c >= 0) *(uint32_t *)&sy[r + 4][c + 4] = NB_LD32(F.rec[0] + (size_t)(gy + r) * W + gx + c);
    }
    for (int i = lane; i < 72; i += 64) {
        const int pl = i / 36, j = i % 36, r = j / 3 - 4, c = (j % 3) * 4 - 4;
        if (own && r >= 0 && c >= 0) *(uint32_t *)&sc[pl][r + 4][c + 4] = lds4(own + 256 + r * 16 + pl * 8 + c);
        else if (cgy + r >= 0 && cgx + c >= 0) *(uint32_t *)&sc[pl][r + 4][c + 4] = NB_LD32((pl ? F.rec[2] : F.rec[1]) + (size_t)(cgy + r) * CW + cgx + c);
    }
    /* boundary strengths */
    const int type = Lo ? Lo->i_type : (int)NB_LD8(&F.mb_type[xy]), qp = F.qp;
    const int qp_thresh = 15 - (F.chroma_qp_offset > 0 ? F.chroma_qp_offset : 0);
    const int edge_end = (type == PCAMV_P_SKIP || qp <= qp_thresh) ? 1 : 4;
    const int no_sub8x8 = type != PCAMV_P_8x8 || !(F.inter & PCAMV_ANALYSE_PSUB8x8);
    if (lane < 32) {
        const int dir = lane >> 4, edge = (lane >> 2) & 3, i = lane & 3;
        int bs = 0;
        const bool on = edge < edge_end && !(edge == 0 && (dir ? my == 0 : mx == 0));
        if (on) {
            const int x = dir == 0 ? edge : i, y = dir == 0 ? i : edge;
            const int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
            const int nxy = edge ? xy : (dir ? xy - F.mb_w : xy - 1);
            const int bi = (x & 1) + 2 * (y & 1) + 4 * (x >> 1) + 8 * (y >> 1), bn = (xn & 1) + 2 * (yn & 1) + 4 * (xn >> 1) + 8 * (yn >> 1);
            const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w;
            const int fx = 4 * mx + x, fy = 4 * my + y, fxn = dir == 0 ? fx - 1 : fx, fyn = dir == 0 ? fy : fy - 1;
            /* both sides of the edge: flags, motion, reference -- this macroblock's from LDS when it has just been made
             * here, the rest in one round of loads (not one per test) */
            const bool nb_local = Lo && edge;
            const int c8a = SCAN8_0 + x + 8 * y, c8b = SCAN8_0 + xn + 8 * yn;
            const unsigned nz_a = Lo ? (unsigned)Lo->nnz_mask : (unsigned)NB_LD16(&F.nnz[xy]);
            const unsigned nz_b = nb_local ? (unsigned)Lo->nnz_mask : (unsigned)NB_LD16(&F.nnz[nxy]);
            const uint32_t wa = Lo ? NB_PACK16(Lo->cmv[c8a][0], Lo->cmv[c8a][1]) : NB_LD32(F.mv + 2 * (fy * s4 + fx));
            const uint32_t wb = nb_local ? NB_PACK16(Lo->cmv[c8b][0], Lo->cmv[c8b][1]) : NB_LD32(F.mv + 2 * (fyn * s4 + fxn));
            const int ra = Lo ? (int)Lo->cref[c8a] : (int)NB_LD8(&F.ref8[(fy >> 1) * s8 + (fx >> 1)]);
            const int rb = nb_local ? (int)Lo->cref[c8b] : (int)NB_LD8(&F.ref8[(fyn >> 1) * s8 + (fxn >> 1)]);
            if (((nz_a >> bi) & 1) || ((nz_b >> bn) & 1)) bs = 2;
            else if (!(edge & no_sub8x8)) {
                const int a0 = (int16_t)(wa & 0xffff), a1 = (int16_t)(wa >> 16), b0 = (int16_t)(wb & 0xffff), b1 = (int16_t)(wb >> 16);
                if (ra != rb || iabs(a0 - b0) >= 4 || iabs(a1 - b1) >= 4) bs = 1;
                bs |= 0x10;              /* marks "decided by the motion test" for the copy rule below */
            }
        }
        sbs[dir][edge][i] = (uint8_t)bs;
    }
    __syncthreads();
    {   /* frame.c:735-737: inside an 8x8 that cannot be split, the odd 4-pixel group repeats its left/upper
         * neighbour's strength unless that one is 2 */
        const int dir = (lane >> 4) & 1, edge = (lane >> 2) & 3, i = lane & 3;
        int bs = sbs[dir][edge][i];
        const int prev = i ? sbs[dir][edge][i - 1] & 0xf : 0;
        __syncthreads();
        if (lane < 32) {
            if ((bs & 0x10) && (i & no_sub8x8) && prev != 2) bs = prev;
            sbs[dir][edge][i] = (uint8_t)(bs & 0xf);
        }
    }
    __syncthreads();
    const int qpc = F.chroma_qp;
    const int alpha = dbk_alpha_dev[qp], beta = dbk_beta_dev[qp], calpha = dbk_alpha_dev[qpc], cbeta = dbk_beta_dev[qpc];
    /* tc0 of the three strengths, looked up once (a per-edge table load would sit on the chain of eight dependent edges) */
    const int tl1 = dbk_tc0_dev[qp][0], tl2 = dbk_tc0_dev[qp][1], tl3 = dbk_tc0_dev[qp][2];
    const int tc1 = dbk_tc0_dev[qpc][0], tc2 = dbk_tc0_dev[qpc][1], tc3 = dbk_tc0_dev[qpc][2];
    for (int dir = 0; dir < 2; dir++)
        for (int edge = 0; edge < 4; edge++) {
            const uint32_t any = *(const uint32_t *)sbs[dir][edge];
            if (any) {
                if (lane < 16 && alpha && beta) {
                    const int bs = sbs[dir][edge][lane >> 2];
                    if (bs) {
                        const int tc0 = bs == 1 ? tl1 : bs == 2 ? tl2 : tl3;
                        uint8_t *q = dir == 0 ? &sy[lane + 4][4 * edge + 4] : &sy[4 * edge + 4][lane + 4];
                        const int xs = dir == 0 ? 1 : 24;
                        const int p2 = q[-3 * xs], p1 = q[-2 * xs], p0 = q[-xs], q0 = q[0], q1 = q[xs], q2 = q[2 * xs];
                        if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
                            int tc = tc0;
                            if (iabs(p2 - p0) < beta) { q[-2 * xs] = (uint8_t)(p1 + clip3i(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0)); tc++; }
                            if (iabs(q2 - q0) < beta) { q[xs] = (uint8_t)(q1 + clip3i(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0)); tc++; }
                            const int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
                            q[-xs] = (uint8_t)clip3i(p0 + delta, 0, 255); q[0] = (uint8_t)clip3i(q0 - delta, 0, 255);
                        }
                    }
                } else if (lane >= 16 && lane < 32 && !(edge & 1) && calpha && cbeta) {
                    const int pl = (lane - 16) >> 3, l = (lane - 16) & 7, bs = sbs[dir][edge][l >> 1];
                    if (bs) {
                        const int tc = (bs == 1 ? tc1 : bs == 2 ? tc2 : tc3) + 1;
                        uint8_t *q = dir == 0 ? &sc[pl][l + 4][2 * edge + 4] : &sc[pl][2 * edge + 4][l + 4];
                        const int xs = dir == 0 ? 1 : 16;
                        const int p1 = q[-2 * xs], p0 = q[-xs], q0 = q[0], q1 = q[xs];
                        if (iabs(p0 - q0) < calpha && iabs(p1 - p0) < cbeta && iabs(q1 - q0) < cbeta) {
                            const int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
                            q[-xs] = (uint8_t)clip3i(p0 + delta, 0, 255); q[0] = (uint8_t)clip3i(q0 - delta, 0, 255);
                        }
                    }
                }
            }
            __syncthreads();
        }
    /* write back: the macroblock, the 4 columns left of it (rows 0..15), the 4 rows above it (cols 0..15) */
    { const int r = lane >> 2, c = (lane & 3) * 4;
      NB_ST32(F.rec[0] + (size_t)(gy + r) * W + gx + c, *(const uint32_t *)&sy[r + 4][c + 4]); }
    if (lane < 16 && mx > 0) NB_ST32(F.rec[0] + (size_t)(gy + lane) * W + gx - 4, *(const uint32_t *)&sy[lane + 4][0]);
    if (lane >= 16 && lane < 32 && my > 0) { const int r = (lane - 16) >> 2, c = ((lane - 16) & 3) * 4;
      NB_ST32(F.rec[0] + (size_t)(gy - 4 + r) * W + gx + c, *(const uint32_t *)&sy[r][c + 4]); }
    if (lane >= 32) {
        const int pl = (lane - 32) >> 4, j = (lane - 32) & 15, r = j >> 1, c = (j & 1) * 4;
        uint8_t *dst = pl ? F.rec[2] : F.rec[1];
        NB_ST32(dst + (size_t)(cgy + r) * CW + cgx + c, *(const uint32_t *)&sc[pl][r + 4][c + 4]);
    }
    __syncthreads();
    if (lane < 16 && mx > 0) { const int pl = lane >> 3, r = lane & 7; uint8_t *dst = pl ? F.rec[2] : F.rec[1];
      NB_ST32(dst + (size_t)(cgy + r) * CW + cgx - 4, *(const uint32_t *)&sc[pl][r + 4][0]); }
    if (lane >= 16 && lane < 32 && my > 0) { const int pl = (lane - 16) >> 3, j = (lane - 16) & 7, r = j >> 1, c = (j & 1) * 4; uint8_t *dst = pl ? F.rec[2] : F.rec[1];
      NB_ST32(dst + (size_t)(cgy - 4 + r) * CW + cgx + c, *(const uint32_t *)&sc[pl][r][c + 4]); }
}
/* ------------------------------------------------------------------ the second pass of a RUN of macroblocks of a row (k_pass2_deblock_flow)
 * A task of the second-pass kernel is up to eight macroblocks of a row.  Taken one by one, each cost two or three memory round trips (record,
 * pixels, the filter's borders), fetched 4 KB for its 0.6 KB (a macroblock's sixteen 16-byte rows are sixteen cache lines, which its seven
 * neighbours in the run fetch again) and stored its rows as partial lines.  Here the run is ONE tile in LDS: its pixels with the four
 * rows above and the four columns to the left (rows of up to 132 bytes: whole lines), the eight records and the neighbours' side of the
 * outer edges come in one round trip; the macroblocks are then reconstructed (where the embedding changed them) and filtered in place, the
 * left neighbour's side of an edge handed from one to the next in LDS; the tile goes back in rows. */
#define P2_TW 144        /* tile row pitch, luma: columns -4 .. 127 at [c + 4] */
#define P2_CW 80         /* chroma: columns -4 .. 63 at [c + 4] */
struct P2Unit {
    uint8_t ty[20][P2_TW];              /* rows -4 .. 15 at [r + 4] */
    uint8_t tc[2][12][P2_CW];
    pcamv_mb_t rec[8];
    int car_base[8], mbflip[8], nnz1[8];
    unsigned t_nnz[8]; uint32_t t_mv[8][4]; int t_ref[8][4];      /* the upper neighbours' bottom row of 4x4 blocks */
    unsigned l_nnz; uint32_t l_mv[4]; int l_ref[4];                 /* the left neighbour's right column: of the run's first macroblock from memory, then handed on */
    uint8_t sbs[2][4][4];
};
#define P2_LSLOTS (20 * 33)             /* luma dwords of the tile: row r = i / 33 - 4, column c = 4 * (i % 33) - 4 */
#define P2_CSLOTS (2 * 12 * 17)         /* chroma: plane i / 204, row (i % 204) / 17 - 4, column 4 * (i % 17) - 4 */
__device__ __forceinline__ bool p2_slot(const FrameDev &F, int i, bool chroma, int x0, int y, int n, int *pl, int *r, int *c, size_t *goff)
{
    if (!chroma) {
        if (i >= P2_LSLOTS) return false;
        *pl = 0; *r = i / 33 - 4; *c = 4 * (i % 33) - 4;
        const int gx = 16 * x0 + *c, gy = 16 * y + *r;
        if (gx < 0 || gy < 0 || *c >= 16 * n) return false;
        *goff = (size_t)gy * F.w + gx;
    } else {
        if (i >= P2_CSLOTS) return false;
        const int j = i % 204;
        *pl = 1 + i / 204; *r = j / 17 - 4; *c = 4 * (j % 17) - 4;
        const int gx = 8 * x0 + *c, gy = 8 * y + *r;
        if (gx < 0 || gy < 0 || *c >= 8 * n) return false;
        *goff = (size_t)gy * (F.w >> 1) + gx;
    }
    return true;
}
__device__ __forceinline__ void p2_unit_load(const FrameDev &F, P2Unit *U, int x0, int y, int n)
{
    const int lane = LANE(), xy0 = y * F.mb_w + x0;
    /* every load first, then the stores to LDS: one round trip for the run */
    uint32_t vl[11], vc[7], vr[8];
#pragma unroll
    for (int t = 0; t < 11; t++) {
        int pl, r, c; size_t o;
        vl[t] = p2_slot(F, lane + 64 * t, false, x0, y, n, &pl, &r, &c, &o) ? NB_LD32(F.rec[0] + o) : 0u;
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        int pl, r, c; size_t o;
        vc[t] = 0u;
        if (p2_slot(F, lane + 64 * t, true, x0, y, n, &pl, &r, &c, &o)) vc[t] = NB_LD32((pl == 2 ? F.rec[2] : F.rec[1]) + o);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int i = lane + 64 * t, k = i / 59, w = i - 59 * k;
        vr[t] = (i < 8 * 59 && k < n) ? ((const uint32_t *)&F.rec_mb[xy0 + k])[w] : 0u;
    }
    int cb = 0, mf = 1, n1 = 0;
    if (lane < n) { cb = F.car_base ? F.car_base[xy0 + lane] : 0; mf = F.mbflip ? (int)F.mbflip[xy0 + lane] : 1; n1 = (int)F.nnz[xy0 + lane]; }
    unsigned tn = 0, ln = 0; uint32_t tm = 0, lm = 0; int tr = 0, lr = 0;
    const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w;
    if (lane < 4 * n && y > 0) {          /* lane = 4 k + j: block j of the bottom row of the macroblock above macroblock k */
        const int k = lane >> 2, j = lane & 3, fx = 4 * (x0 + k) + j, fy = 4 * y - 1;
        tn = (unsigned)NB_LD16(&F.nnz[xy0 + k - F.mb_w]); tm = NB_LD32(F.mv + 2 * (fy * s4 + fx)); tr = (int)NB_LD8(&F.ref8[(fy >> 1) * s8 + (fx >> 1)]);
    }
    if (lane >= 32 && lane < 36 && x0 > 0) {      /* block (3, j) of the macroblock left of the run */
        const int j = lane - 32, fx = 4 * x0 - 1, fy = 4 * y + j;
        ln = (unsigned)NB_LD16(&F.nnz[xy0 - 1]); lm = NB_LD32(F.mv + 2 * (fy * s4 + fx)); lr = (int)NB_LD8(&F.ref8[(fy >> 1) * s8 + (fx >> 1)]);
    }
    PCAMV_WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < 11; t++) {
        int pl, r, c; size_t o;
        if (p2_slot(F, lane + 64 * t, false, x0, y, n, &pl, &r, &c, &o)) *(uint32_t *)&U->ty[r + 4][c + 4] = vl[t];
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        int pl, r, c; size_t o;
        if (p2_slot(F, lane + 64 * t, true, x0, y, n, &pl, &r, &c, &o)) *(uint32_t *)&U->tc[pl - 1][r + 4][c + 4] = vc[t];
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int i = lane + 64 * t;
        if (i < 8 * 59) ((uint32_t *)U->rec)[i] = vr[t];
    }
    if (lane < 8) { U->car_base[lane] = cb; U->mbflip[lane] = mf; U->nnz1[lane] = n1; }
    if (lane < 32) { U->t_mv[lane >> 2][lane & 3] = tm; U->t_ref[lane >> 2][lane & 3] = tr; if ((lane & 3) == 0) U->t_nnz[lane >> 2] = tn; }
    if (lane >= 32 && lane < 36) { U->l_mv[lane - 32] = lm; U->l_ref[lane - 32] = lr; if (lane == 32) U->l_nnz = ln; }
    PCAMV_WAVE_SYNC();
}
/* a reconstructed macroblock (L->pred) into its place in the tile */
__device__ __forceinline__ void p2_put_mb(P2Unit *U, const MBLocal *L, int k)
{
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
    *(uint32_t *)&U->ty[(lane >> 2) + 4][16 * k + 4 + 4 * (lane & 3)] = lds4(L->pred + (lane >> 2) * 16 + (lane & 3) * 4);
    if (lane < 32) *(uint32_t *)&U->tc[lane >> 4][((lane & 15) >> 1) + 4][8 * k + 4 + 4 * (lane & 1)] = lds4(L->pred + 256 + ((lane & 15) >> 1) * 16 + (lane >> 4) * 8 + (lane & 1) * 4);
    PCAMV_WAVE_SYNC();
}
/* the loop filter of macroblock k of the run, in the tile (what mbk_deblock does in its staging area; same strengths, same arithmetic) */
__device__ __forceinline__ void mbk_deblock_unit(const FrameDev &F, P2Unit *U, const MBLocal *Lo, int k, int mx, int my)
{
    uint8_t (*sbs)[4][4] = U->sbs;
    const int lane = LANE();
    const int type = Lo->i_type, qp = F.qp;
    const int qp_thresh = 15 - (F.chroma_qp_offset > 0 ? F.chroma_qp_offset : 0);
    const int edge_end = (type == PCAMV_P_SKIP || qp <= qp_thresh) ? 1 : 4;
    const int no_sub8x8 = type != PCAMV_P_8x8 || !(F.inter & PCAMV_ANALYSE_PSUB8x8);
    PCAMV_WAVE_SYNC();
    if (lane < 32) {
        const int dir = lane >> 4, edge = (lane >> 2) & 3, i = lane & 3;
        int bs = 0;
        const bool on = edge < edge_end && !(edge == 0 && (dir ? my == 0 : mx == 0));
        if (on) {
            const int x = dir == 0 ? edge : i, y = dir == 0 ? i : edge;
            const int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
            const int bi = (x & 1) + 2 * (y & 1) + 4 * (x >> 1) + 8 * (y >> 1), bn = (xn & 1) + 2 * (yn & 1) + 4 * (xn >> 1) + 8 * (yn >> 1);
            const int c8a = SCAN8_0 + x + 8 * y, c8b = SCAN8_0 + xn + 8 * yn;
            const unsigned nz_a = (unsigned)Lo->nnz_mask;
            const unsigned nz_b = edge ? (unsigned)Lo->nnz_mask : dir ? U->t_nnz[k] : U->l_nnz;
            const uint32_t wa = NB_PACK16(Lo->cmv[c8a][0], Lo->cmv[c8a][1]);
            const uint32_t wb = edge ? NB_PACK16(Lo->cmv[c8b][0], Lo->cmv[c8b][1]) : dir ? U->t_mv[k][i] : U->l_mv[i];
            const int ra = (int)Lo->cref[c8a];
            const int rb = edge ? (int)Lo->cref[c8b] : dir ? U->t_ref[k][i] : U->l_ref[i];
            if (((nz_a >> bi) & 1) || ((nz_b >> bn) & 1)) bs = 2;
            else if (!(edge & no_sub8x8)) {
                const int a0 = (int16_t)(wa & 0xffff), a1 = (int16_t)(wa >> 16), b0 = (int16_t)(wb & 0xffff), b1 = (int16_t)(wb >> 16);
                if (ra != rb || iabs(a0 - b0) >= 4 || iabs(a1 - b1) >= 4) bs = 1;
                bs |= 0x10;
            }
        }
        sbs[dir][edge][i] = (uint8_t)bs;
    }
    __syncthreads();
    {
        const int dir = (lane >> 4) & 1, edge = (lane >> 2) & 3, i = lane & 3;
        int bs = sbs[dir][edge][i];
        const int prev = i ? sbs[dir][edge][i - 1] & 0xf : 0;
        __syncthreads();
        if (lane < 32) {
            if ((bs & 0x10) && (i & no_sub8x8) && prev != 2) bs = prev;
            sbs[dir][edge][i] = (uint8_t)(bs & 0xf);
        }
    }
    __syncthreads();
    const int qpc = F.chroma_qp;
    const int alpha = dbk_alpha_dev[qp], beta = dbk_beta_dev[qp], calpha = dbk_alpha_dev[qpc], cbeta = dbk_beta_dev[qpc];
    const int tl1 = dbk_tc0_dev[qp][0], tl2 = dbk_tc0_dev[qp][1], tl3 = dbk_tc0_dev[qp][2];
    const int tc1 = dbk_tc0_dev[qpc][0], tc2 = dbk_tc0_dev[qpc][1], tc3 = dbk_tc0_dev[qpc][2];
    /* luma lines in lanes 0..15, the chroma lines of the even edges in lanes 16..31 (plane, line), ONE instruction stream for both: the chroma
     * filter is the luma one without the second-neighbour terms and with tc = tc0 + 1 (deblock_chroma_c vs deblock_luma_c, common/frame.c) */
    const bool is_c = lane >= 16;
    const int cpl = (lane - 16) >> 3, cl = (lane - 16) & 7;
    const int f_alpha = is_c ? calpha : alpha, f_beta = is_c ? cbeta : beta;
    for (int dir = 0; dir < 2; dir++)
        for (int edge = 0; edge < 4; edge++) {
            const uint32_t any = *(const uint32_t *)sbs[dir][edge];
            if (any) {
                if (lane < 32 && f_alpha && f_beta && !(is_c && (edge & 1))) {
                    const int bs = sbs[dir][edge][is_c ? cl >> 1 : lane >> 2];
                    if (bs) {
                        const int tc0 = is_c ? (bs == 1 ? tc1 : bs == 2 ? tc2 : tc3) : (bs == 1 ? tl1 : bs == 2 ? tl2 : tl3);
                        uint8_t *q = is_c ? (dir == 0 ? &U->tc[cpl][cl + 4][8 * k + 2 * edge + 4] : &U->tc[cpl][2 * edge + 4][8 * k + cl + 4])
                                          : (dir == 0 ? &U->ty[lane + 4][16 * k + 4 * edge + 4] : &U->ty[4 * edge + 4][16 * k + lane + 4]);
                        const int xs = dir == 0 ? 1 : is_c ? P2_CW : P2_TW;
                        const int p2 = q[-3 * xs], p1 = q[-2 * xs], p0 = q[-xs], q0 = q[0], q1 = q[xs], q2 = q[2 * xs];
                        if (iabs(p0 - q0) < f_alpha && iabs(p1 - p0) < f_beta && iabs(q1 - q0) < f_beta) {
                            const bool ap = !is_c && iabs(p2 - p0) < f_beta, aq = !is_c && iabs(q2 - q0) < f_beta;
                            const int tc = is_c ? tc0 + 1 : tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
                            if (ap) q[-2 * xs] = (uint8_t)(p1 + clip3i(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0));
                            if (aq) q[xs] = (uint8_t)(q1 + clip3i(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0));
                            const int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
                            q[-xs] = (uint8_t)clip3i(p0 + delta, 0, 255); q[0] = (uint8_t)clip3i(q0 - delta, 0, 255);
                        }
                    }
                }
            }
            __syncthreads();
        }
    /* this macroblock's right column of 4x4 blocks is the next one's left neighbour */
    if (lane < 4) { const int c8 = SCAN8_0 + 3 + 8 * lane; U->l_mv[lane] = NB_PACK16(Lo->cmv[c8][0], Lo->cmv[c8][1]); U->l_ref[lane] = (int)Lo->cref[c8]; }
    if (lane == 0) U->l_nnz = (unsigned)Lo->nnz_mask;
    PCAMV_WAVE_SYNC();
}
/* the tile back to the frame: the run's rows 0..15 with the four columns left of it (the left neighbour's, touched by the first
 * macroblock's left edge), and the four rows above it */
__device__ __forceinline__ void p2_unit_store(const FrameDev &F, P2Unit *U, int x0, int y, int n)
{
    const int lane = LANE();
    PCAMV_WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < 11; t++) {
        int pl, r, c; size_t o;
        if (p2_slot(F, lane + 64 * t, false, x0, y, n, &pl, &r, &c, &o) && (r < 0 ? c >= 0 : true)) NB_ST32(F.rec[0] + o, *(const uint32_t *)&U->ty[r + 4][c + 4]);
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        int pl, r, c; size_t o;
        if (p2_slot(F, lane + 64 * t, true, x0, y, n, &pl, &r, &c, &o) && (r < 0 ? c >= 0 : true)) NB_ST32((pl == 2 ? F.rec[2] : F.rec[1]) + o, *(const uint32_t *)&U->tc[pl - 1][r + 4][c + 4]);
    }
}
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_deblock_diag(const FrameDev *__restrict__ Fs, int d)
{
    __shared__ DeblockLDS D;
    const FrameDev F = Fs[blockIdx.y];
    int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    const int my = y_lo + (int)blockIdx.x, mx = d - 2 * my;
    if (my >= F.mb_h || mx < 0 || mx >= F.mb_w) return;
    mbk_deblock(F, &D, mx, my);
}
#endif

/* both stages of one anti-diagonal in one launch: the filter of (x,y) only needs the pass-2 reconstruction of
 * (x,y) itself and the filtered neighbours of earlier diagonals, and only modifies macroblocks of earlier diagonals */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_pass2_deblock_diag(const FrameDev *__restrict__ Fs, int d)
{
    __shared__ MBLocal L;
    __shared__ DeblockLDS D;
    const FrameDev F = Fs[blockIdx.y];
    int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    const int y = y_lo + (int)blockIdx.x, x = d - 2 * y;
    if (y >= F.mb_h || x < 0 || x >= F.mb_w) return;
    mbk_pass2(F, &L, x, y);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        /* the reconstruction just stored is read back by the filter */
    __syncthreads();
    mbk_deblock(F, &D, x, y);
}
#endif

/* ------------------------------------------------------------------ dataflow scheduling of the analysis
 * One persistent launch per frame step instead of one launch per anti-diagonal: macroblock (x,y) of a
 * GOP becomes ready when (x-1,y) and (x+1,y-1) [or (x,y-1) at the right edge] are done; ready
 * macroblocks of every GOP in flight go through ONE append-only queue.  A wave pops the next index,
 * waits for that entry to be published, runs the search, publishes the motion the neighbours need
 * (agent-scope release), decrements its two successors' dependency counters (the one that reaches 0
 * is appended to the queue), and then -- off the critical path -- does the macroblock's RCA costs and
 * pass-1 reconstruction from the state it just produced.  Nothing depends on dispatch order, on
 * residency or on workgroup->XCD placement: an entry index is only waited for after it was handed out,
 * entries are appended by waves that are running, and the dependency graph always has a ready node
 * until everything is done.  Spins are bounded; a timeout raises ctr[2] and every wave drains. */
struct FlowDev {
    unsigned *ctr;            /* FLOW_HEAD(q) pop index / FLOW_TAIL(q) append index of queue q, FLOW_ERR error flag -- every
                               * counter in a 128-byte line of its own: they are the hottest addresses of the launch, and
                               * with all sixteen in one line every pop and append of the whole chip serialised on it
                               * (23 ns per macroblock: the entire cost of the pass-2 kernel, and a floor under the search) */
    unsigned *queue;          /* total entries, queue x at [qbase[x], qbase[x] + qcount[x]); 0 = not yet published, else (gop << 16 | mb_xy) + 1 */
    int *dep;                 /* [n_gop * n_mb] dependencies still open */
    unsigned total, spin_limit;
    unsigned qbase[8], qcount[8];
    int n_gop, n_mb, mb_w, mb_h, fused, nq;      /* nq = 8: GOP g lives in queue g & 7 (XCD affinity); nq = 1: one queue */
    int unit;                 /* macroblocks of a row per task (second pass only; 1 by default); mb_w / n_mb above are in tasks */
    int raster;               /* --subme >= 6 with CABAC: the slice's context states chain the macroblocks of a frame in raster order
                               * (encoder.c:1900-1927, rdo.c:62), so a macroblock's only predecessor is the one coded before it */
    int spec;                 /* raster chains handed on speculatively (mbk_search_spec below); the launcher picks the kernel instance */
    unsigned *rdone;          /* [n_gop * FLOW_RDONE_STRIDE] per chain: macroblocks of the frame whose FINAL state is published */
};
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define FLOW_HEAD(q) (64 * (q))
#define FLOW_TAIL(q) (64 * (q) + 32)
#define FLOW_ERR 512
#define FLOW_CTR_WORDS 576
#define FLOW_RDONE_STRIDE 32          /* words between two chains' counters: a 128-byte line each */
#define FLOW_SPEC_AHEAD 4             /* a macroblock is handed on only once the one FLOW_SPEC_AHEAD before it is final */
#define FLOW_SPEC_MIN_MBW 8           /* (so that its top / top-right neighbours, mb_w - 1 .. mb_w + 1 back, always are) */

#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(256) k_flow_init(FlowDev fl)
{
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= fl.total) return;
    int xy = (int)(i % (unsigned)fl.n_mb), x = xy % fl.mb_w, y = xy / fl.mb_w;
    fl.dep[i] = fl.raster ? (xy > 0) : (x > 0) + (y > 0);
    /* word i of the queue array is entry k of queue q; the first ngop(q) entries of each queue start out
     * published: macroblock 0 of the GOPs k * nq + q */
    int q = 0;
    while (q + 1 < fl.nq && i >= fl.qbase[q + 1]) q++;
    const unsigned k = i - fl.qbase[q], ngop_q = fl.qcount[q] / (unsigned)fl.n_mb;
    fl.queue[i] = k < ngop_q ? ((k * (unsigned)fl.nq + (unsigned)q) << 16) + 1u : 0u;
    if (i < 8) { fl.ctr[FLOW_HEAD(i)] = 0u; fl.ctr[FLOW_TAIL(i)] = fl.qcount[i] / (unsigned)fl.n_mb; }     /* heads 0, tails = GOPs of the queue; the error flag is the host's */
    if (fl.spec && i < (unsigned)fl.n_gop) fl.rdone[FLOW_RDONE_STRIDE * i] = 0u;
}
#endif

__device__ __forceinline__ unsigned flow_bcast(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void flow_done_one(const FlowDev &fl, int q, int slot, unsigned item)
{
    if (__hip_atomic_fetch_sub(&fl.dep[slot], 1, RLX_AGENT) == 1) {
        unsigned t = __hip_atomic_fetch_add(&fl.ctr[FLOW_TAIL(q)], 1u, RLX_AGENT);
        __hip_atomic_store(&fl.queue[fl.qbase[q] + t], item, RLX_AGENT);
    }
}


/* ------------------------------------------------------------------ speculative hand-off along a raster chain
 * With CABAC a frame is ONE chain of macroblocks (the context states), and with few GOPs in flight the chip waits for that
 * chain: ~100 us per macroblock, of which the 16x16 search is a third.  What the NEXT macroblock's searches need from this
 * one is its motion only (the left column of its 4x4 motion field, its 16x16 search result, whether it is skipped) -- the
 * entropy coder's state is first read by the RD stage.  And the motion is almost always the 16x16 result (or the skip
 * prediction).  So a macroblock publishes "16x16, this MV" (or its skip) right after its 16x16 search and hands the chain on;
 * its other searches, its RD stage and the successor's searches then run side by side on different waves.  Exact by
 * construction: before its RD stage every macroblock waits until its predecessor is FINAL (rdone), compares the motion it
 * started from with the final one, and starts over if they differ (so does, in turn, whoever started from what it published);
 * what a macroblock commits is computed from verified inputs only.  A macroblock is handed on only once the macroblock
 * FLOW_SPEC_AHEAD before it is final, which keeps the top neighbours (>= mb_w - 1 back) out of the speculation.
 * Waits are on waves that are running and never wait for a younger macroblock: no cycle; all spins are bounded. */
__device__ __forceinline__ bool flow_wait_rdone(const FlowDev &fl, int g, unsigned need)
{
    if (!need) return true;
    const unsigned *p = fl.rdone + FLOW_RDONE_STRIDE * g;
    for (unsigned spins = 0;; spins++) {
        unsigned v = 0;
        if (LANE() == 0) v = __hip_atomic_load(p, RLX_AGENT);
        if (flow_bcast(v) >= need) return true;
        unsigned bad = 0;
        if ((spins & 255u) == 255u) { if (LANE() == 0) bad = __hip_atomic_load(&fl.ctr[FLOW_ERR], RLX_AGENT); bad = flow_bcast(bad); }
        if (bad || spins >= fl.spin_limit) { if (LANE() == 0) __hip_atomic_store(&fl.ctr[FLOW_ERR], 1u, RLX_AGENT); return false; }
        if (spins < 16) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);
    }
}
/* the left neighbour's motion as this macroblock's searches used it (from the cache and the candidate list: the very values
 * they consumed), and as it is now in memory; equal = the searches stand */
__device__ __forceinline__ bool spec_inputs_final(const FrameDev &F, MBLocal *L)
{
    if (!(L->neighbour & NB_LEFT)) return true;
    const int xy = L->mb_xy, s4 = 4 * F.mb_w, b4 = 4 * (L->mb_y * s4 + L->mb_x);
    const uint32_t m0 = NB_LD32(&F.mv[2 * (b4 - 1)]), m1 = NB_LD32(&F.mv[2 * (b4 - 1 + s4)]);
    const uint32_t m2 = NB_LD32(&F.mv[2 * (b4 - 1 + 2 * s4)]), m3 = NB_LD32(&F.mv[2 * (b4 - 1 + 3 * s4)]);
    const uint32_t r = NB_LD32(&F.mvr[2 * (xy - 1)]);
    const int t = NB_LD8(&F.mb_type[xy - 1]);
    PCAMV_WAVE_SYNC();
    const int16_t (*c)[2] = L->cmv;
    bool ok = m0 == NB_PACK16(c[SCAN8_0 - 1][0], c[SCAN8_0 - 1][1]) && m1 == NB_PACK16(c[SCAN8_0 - 1 + 8][0], c[SCAN8_0 - 1 + 8][1]) &&
              m2 == NB_PACK16(c[SCAN8_0 - 1 + 16][0], c[SCAN8_0 - 1 + 16][1]) && m3 == NB_PACK16(c[SCAN8_0 - 1 + 24][0], c[SCAN8_0 - 1 + 24][1]);
    ok = ok && (t == PCAMV_P_SKIP) == (L->type_left == PCAMV_P_SKIP);
    /* a coded left neighbour's 16x16 result is the first candidate of this macroblock's 16x16 search (predict_mv_ref16x16) */
    if (t != PCAMV_P_SKIP) ok = ok && r == NB_PACK16(L->mvc16[0][0], L->mvc16[0][1]);
    return flow_bcast(ok ? 1u : 0u) != 0u;
}
template <int TESA>
__device__ __forceinline__ bool mbk_search_spec(const FrameDev &F, MBLocal *L, Analysis *a, int mb_x, int mb_y, const FlowDev &fl, int g, unsigned item)
{
    const int xy = mb_y * F.mb_w + mb_x, lane = LANE();
    const int s4 = 4 * F.mb_w, s8 = 2 * F.mb_w, b4 = 4 * (mb_y * s4 + mb_x), b8 = 2 * (mb_y * s8 + mb_x);
    bool handed_on = false;
    int skip;
    for (int round = 0;; round++) {
        mb_load(F, L, mb_x, mb_y, 0, 0);                 /* neighbours' motion + source pixels; nothing of the entropy coder yet */
        skip = analyse_s16<TESA>(F, L, a);
        if (!handed_on) {
            /* what the successor's searches start from: a skipped macroblock's motion is final as it stands (as far as this
             * macroblock's own inputs are), a coded one is announced as 16x16 with the search's result */
            const uint32_t w = skip ? NB_PACK16(L->pskip_mv[0], L->pskip_mv[1]) : NB_PACK16(a->me16x16.mv[0], a->me16x16.mv[1]);
            if (lane < 16) NB_ST32(&F.mv[2 * (b4 + (lane >> 2) * s4 + (lane & 3))], w);
            if (lane == 0) { NB_ST8(&F.mb_type[xy], skip ? PCAMV_P_SKIP : PCAMV_P_L0); NB_ST16(&F.ref8[b8], 0); NB_ST16(&F.ref8[b8 + s8], 0); }
            if (xy + 1 < fl.n_mb) {
                if (!flow_wait_rdone(fl, g, xy + 1 > FLOW_SPEC_AHEAD ? (unsigned)(xy + 1 - FLOW_SPEC_AHEAD) : 0u)) return false;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) flow_done_one(fl, g & (fl.nq - 1), g * fl.n_mb + xy + 1, item + 1u);
            }
            handed_on = true;
        }
        if (!skip) analyse_s_rest<TESA>(F, L, a);
        if (!flow_wait_rdone(fl, g, (unsigned)xy)) return false;         /* the macroblock coded before this one is final */
        if (spec_inputs_final(F, L)) break;
        if (round >= 64) { if (lane == 0) __hip_atomic_store(&fl.ctr[FLOW_ERR], 1u, RLX_AGENT); return false; }     /* (cannot happen: the predecessor is final now) */
    }
    /* RD stage: the entropy coder's neighbourhood, the intra borders and the context states as the predecessor left them */
    {
        MbFetch pf;
        prim_mb_fetch(F, mb_x, mb_y, L->neighbour, 1, pf);
        prim_mb_fetch_store(F, L, 1, pf);
    }
    if (!skip) analyse_decide<TESA>(F, L, a);
    update_cache(L, a);
    mbk_search_finish<TESA>(F, L, a, mb_x, mb_y);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(fl.rdone + FLOW_RDONE_STRIDE * g, (unsigned)(xy + 1), RLX_AGENT);
    return true;
}

#ifndef PCAMV_FLOW_OCC
#define PCAMV_FLOW_OCC 4        /* waves per SIMD the register allocation of the persistent kernel is held to */
#endif
/* the queue protocol, shared by the two persistent kernels; MODE 0: search -> publish -> reconstruction + RCA,
 * MODE 1: pass 2 + loop filter of the macroblock -> publish */
template <int MODE, int TESA>
__device__ __forceinline__ void flow_loop(const FrameDev *__restrict__ Fs, const FlowDev &fl, MBLocal &L, Analysis *Ap, P2Unit *Up)
{
    const int lane = LANE();
    /* home queue = this wave's XCD (speed only: the GOPs of one queue are then searched through one L2
     * instead of being replicated in all eight); a wave whose queue is handed out moves on to the others */
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int home = (int)(xcc & 7u) & (fl.nq - 1), tried = 0;
    /* the queue ticket of the next macroblock is taken before the RCA step of the current one (nothing there depends
     * on other waves), so the atomic's round trip is covered by work instead of being waited for (measured +2 %;
     * reading the queue entry early as well gave nothing more) */
    unsigned ticket = 0;
    bool have_ticket = false;
    PROF_INIT();
    for (;;) {
        const unsigned long long t_pop = PROF_T();
        unsigned idx = ticket;
        if (!have_ticket && lane == 0) idx = __hip_atomic_fetch_add(&fl.ctr[FLOW_HEAD(home)], 1u, RLX_AGENT);
        have_ticket = false;
        idx = flow_bcast(idx);
        if (MODE == 0) PROF_ADD(13, t_pop);
        const unsigned long long t_item = PROF_T();
        if (idx >= fl.qcount[home]) {                      /* this queue is handed out: next one, or done */
            if (++tried >= fl.nq) break;
            home = (home + 1) & (fl.nq - 1);
            continue;
        }
        tried = 0;
        const unsigned slot = fl.qbase[home] + idx;
        unsigned item = 0;
        for (unsigned spins = 0;; spins++) {
            unsigned v = 0;
            if (lane == 0) v = __hip_atomic_load(&fl.queue[slot], RLX_AGENT);
            item = flow_bcast(v);
            if (item) break;
            unsigned bad = 0;
            if ((spins & 255u) == 255u) { if (lane == 0) bad = __hip_atomic_load(&fl.ctr[FLOW_ERR], RLX_AGENT); bad = flow_bcast(bad); }
            if (bad || spins >= fl.spin_limit) break;
            if (spins < 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
        }
        if (!item) { if (lane == 0) __hip_atomic_store(&fl.ctr[FLOW_ERR], 1u, RLX_AGENT); break; }
        if (MODE == 0) PROF_ADD(14, t_item);
        const unsigned long long t_f = PROF_T();
#ifdef PCAMV_FLOW_ACQUIRE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         /* drop this CU's stale L1 lines of the neighbours' motion */
#else
        /* no agent-scope acquire: the only data of other waves read here is the neighbours' motion, and every such
         * load is itself an agent-scope load (NB_LD*, `sc1`) issued after the queue entry was seen -- so this CU's L1
         * keeps its lines of the reference planes instead of losing them once per macroblock and wave */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
        const int g = (int)((item - 1u) >> 16), xy = (int)((item - 1u) & 0xffffu);
        const FrameDev F = Fs[g];
        const int y = xy / fl.mb_w, x = xy - y * fl.mb_w;
        if (MODE == 0) PROF_ADD(15, t_f);
        PROF_ADD(MODE ? 13 : 0, t_pop);
        const unsigned long long t_s = PROF_T();
#ifdef PCAMV_SEARCH_CALL
        if (MODE == 0 && lane == 0) L.fdesc = Fs + g;
#endif
        if (MODE == 0 && (TESA & 4)) { if (!mbk_search_spec<TESA>(F, &L, Ap, x, y, fl, g, item)) break; }
        else if (MODE == 0) mbk_search<TESA>(F, &L, Ap, x, y);
        else {
            const int x0 = fl.unit * x, n = imin(fl.unit, F.mb_w - x0);
            p2_unit_load(F, Up, x0, y, n);
            for (int k = 0; k < n; k++) {
                P2Pre pre;
                pre.r = &Up->rec[k]; pre.base = Up->car_base[k]; pre.any_flip = Up->mbflip[k]; pre.nnz1 = Up->nnz1[k]; pre.drain = k > 0;
                if (mbk_pass2(F, &L, x0 + k, y, 0, &pre)) p2_put_mb(Up, &L, k);
                mbk_deblock_unit(F, Up, &L, k, x0 + k, y);
            }
            p2_unit_store(F, Up, x0, y, n);
        }
        PROF_ADD(MODE ? 14 : 1, t_s);
        const unsigned long long t_p = PROF_T();
        /* publish: the motion the neighbours read was stored write-through (NB_ST*, `sc1`); once this wave's stores
         * have drained, the counters / queue entries may follow -- no agent-scope release (it would write back the
         * XCD's whole dirty L2 once per macroblock: measured 13.3 -> 19.9 M MB/s without it) */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0 && (TESA & 4)) { }                    /* the speculative chain hands on inside mbk_search_spec */
        else if (lane == 0 && fl.raster) {
            if (xy + 1 < fl.n_mb) flow_done_one(fl, g & (fl.nq - 1), g * fl.n_mb + xy + 1, item + 1u);
        } else if (lane == 0) {
            const int base = g * fl.n_mb, q = g & (fl.nq - 1);
            if (x + 1 < fl.mb_w) flow_done_one(fl, q, base + xy + 1, item + 1u);
            if (y + 1 < fl.mb_h) {
                if (x >= 1) flow_done_one(fl, q, base + xy + fl.mb_w - 1, item + (unsigned)fl.mb_w - 1u);
                if (x == fl.mb_w - 1) flow_done_one(fl, q, base + xy + fl.mb_w, item + (unsigned)fl.mb_w);
            }
        }
        PROF_ADD(MODE ? 15 : 2, t_p);
        const unsigned long long t_r = PROF_T();
        if (MODE == 0 && fl.fused) {
            /* (not in raster order: a frame is then one chain, its next macroblock is the only work it has, and an entry bound to a
             * ticket whose wave is still busy with this RCA step waits for it while free waves wait for later entries) */
            if (!fl.raster) {
                if (lane == 0) ticket = __hip_atomic_fetch_add(&fl.ctr[FLOW_HEAD(home)], 1u, RLX_AGENT);
                have_ticket = true;
            }
            mbk_rca_encode(F, &L, Ap, xy, 1, (TESA & 2) && F.b_mbrd);
        }
        PROF_ADD(3, t_r);
        if (MODE == 0) PROF_ADD(4, t_pop);
    }
    PROF_FLUSH();
}

#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64, PCAMV_FLOW_OCC) k_analyse_flow(const FrameDev *__restrict__ Fs, FlowDev fl)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    flow_loop<0, 0>(Fs, fl, L, &A, nullptr);
}
#endif
/* The same kernel with --me tesa compiled in (pcamv_logic.h: the search functions are templates on it) lives in a
 * translation unit of its own, csrc/pcamv_tesa.hip, built in parallel with this one; the library calls it through
 * this launcher. */
#ifdef PCAMV_TESA_TU
static __global__ void __launch_bounds__(64, PCAMV_FLOW_OCC) k_analyse_flow_tesa(const FrameDev *__restrict__ Fs, FlowDev fl)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    flow_loop<0, 1>(Fs, fl, L, &A, nullptr);
}
#endif
void pcamv_launch_flow_tesa(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
/* ... and so does the instance with the RD mode decision of --subme 6 / 7 (csrc/pcamv_rd.hip) */
#ifdef PCAMV_RD_TU
/* register budget: PCAMV_RD_OCC waves per SIMD, set by the translation unit (pcamv_rd.hip: 4, pcamv_rd_lo.hip: 1) */
#ifndef PCAMV_RD_OCC
#define PCAMV_RD_OCC 4
#endif
#ifndef PCAMV_RD_VARIANT
#define PCAMV_RD_VARIANT 2        /* variant mask of the control code: 2 = RD mode decision, 2 | 4 = ... with the speculative raster chain */
#endif
static __global__ void __launch_bounds__(64, PCAMV_RD_OCC) k_analyse_flow_rd(const FrameDev *__restrict__ Fs, FlowDev fl)
{
    __shared__ MBLocal L;
    __shared__ Analysis A;
    flow_loop<0, PCAMV_RD_VARIANT>(Fs, fl, L, &A, nullptr);
}
#endif
void pcamv_launch_flow_rd(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu(void);
void pcamv_launch_flow_rd_lo(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu_lo(void);
void pcamv_launch_flow_rd_spec(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu_spec(void);
void pcamv_launch_flow_rd_tesa(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu_tesa(void);
void pcamv_launch_flow_rd_spec2(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu_spec2(void);
void pcamv_launch_flow_rd_spec4(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl);
int pcamv_flow_rd_waves_per_cu_spec4(void);

/* pass 2 + loop filter through the same queue: the tasks are short (~5 us), which only works because the hand-off
 * costs no cache maintenance -- final motion and reconstructed pixels are stored write-through (NB_ST*) and the
 * filter reads its neighbourhood with agent-scope loads (NB_LD*).  (With an agent-scope release + acquire per
 * macroblock this was slower than one launch per anti-diagonal: 245 vs 176 ms per closed-loop step at G=256.) */
#ifndef PCAMV_PASS2_OCC
#define PCAMV_PASS2_OCC 4         /* waves per SIMD the second-pass kernel's registers are held to (its LDS -- the tile of a run of macroblocks -- allows four) */
#endif
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64, PCAMV_PASS2_OCC) k_pass2_deblock_flow(const FrameDev *__restrict__ Fs, FlowDev fl)
{
    /* only the head of the per-macroblock storage (PCAMV_PASS2_LDS: the fields the second pass touches come first in MBLocal): 4.3 instead
     * of 8.9 KB per wave with the filter's staging area, so the CU holds the six waves per SIMD the kernel's 83 VGPRs allow -- it
     * waits for memory three quarters of its time, more waves in flight is what it can use */
    __shared__ __attribute__((aligned(16))) uint8_t Lraw[PCAMV_PASS2_LDS];
    __shared__ __attribute__((aligned(16))) P2Unit U;
    flow_loop<1, 0>(Fs, fl, *reinterpret_cast<MBLocal *>(Lraw), nullptr, &U);
}
#endif

/* block-cost probe: the pixel metrics of a1/a2/a5/a6 (SAD, SATD, qpel fetch, chroma MC) at arbitrary
 * positions, for checkasm-style parity tests through the C ABI.  req = {mb_x,mb_y,ip,xoff,yoff,mx,my,satd} */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_block_costs(const FrameDev *__restrict__ Fs, const int *__restrict__ req, int *__restrict__ out)
{
    __shared__ MBLocal L;
    const FrameDev F = Fs[0];
    const int *r = req + 8 * blockIdx.x;
    L.mb_x = r[0]; L.mb_y = r[1]; L.mb_xy = r[1] * F.mb_w + r[0];
    prim_load_fenc(F, &L);
    PCAMV_WAVE_SYNC();
    const int mflag = (r[7] & 1 ? EV_SATD : 0) | EV_NOMV;
    if (r[7] & 2) {     /* batch mode: 4 candidates around (mx,my) in one list; answer = 3 of them */
        if (LANE() == 0) {
            L.cxy[0] = CAND_PACK(r[5], r[6]); L.cxy[1] = CAND_PACK(r[5] + 1, r[6] - 1);
            L.cxy[2] = CAND_PACK(r[5] - 2, r[6] + 3); L.cxy[3] = CAND_PACK(r[5] + 3, r[6] + 2);
        }
        prim_eval_list(F, &L, L.fenc, r[2], r[3], r[4], 4, mflag, 0, 0);
        if (LANE() == 0) { out[3 * blockIdx.x] = L.ccost[1]; out[3 * blockIdx.x + 1] = L.ccost[2]; out[3 * blockIdx.x + 2] = L.ccost[3]; }
        return;
    }
    if (LANE() == 0) { L.cxy[0] = CAND_PACK(r[5], r[6]); L.ccost[64] = 0; L.ccost[128] = 0; }
    prim_eval_list(F, &L, L.fenc, r[2], r[3], r[4], 1, mflag | EV_CHROMA | EV_PROBE, 0, 0);
    if (LANE() == 0) { out[3 * blockIdx.x] = L.ccost[0]; out[3 * blockIdx.x + 1] = L.ccost[64]; out[3 * blockIdx.x + 2] = L.ccost[128]; }
}
#endif

/* probe of the RD stage's pixel metrics and intra predictors (SURVEY a3 + the intra SATD analysis of --subme >= 6) on caller-supplied
 * pixels, for parity tests against reference-minted vectors through the C ABI.  One request = 1024 bytes: source macroblock
 * (fenc layout: Y 16x16, then U | V 8x8 side by side, stride 16), a second macroblock in the same layout ("reconstruction"),
 * intra borders top[3][28] ([c][3] = top left, [c][4 + x]) and left[3][16], int32 avail (bit 0 left, bit 1 top) at byte 900.
 * out[32]: 0 ssd of the macroblock without the psy term (x264_pixel_ssd 16x16 + 2 x 8x8, pixel.c:71-96), 1 the same with it
 * (ssd_mb, rdo.c:106-137), 2 / 3 hadamard_ac 16x16 of the second block (pixel.c:306-358; 4x4 / 8x8 energies), 4 / 5 the source's
 * psy-RD energies (x264_mb_cache_fenc_satd, analyse.c:522-549: satd / sa8d sums), 6..9 intra 16x16 costs V, H, DC (the variant avail
 * allows), P (common/predict.c + satd or, at subme 1, sad), 10..13 intra chroma DC, H, V, P over both planes, 14..25 the twelve 4x4
 * modes of block 0 (I4_V .. I4_DC_128); an unavailable mode answers PCAMV_COST_MAX */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_rd_probe(const FrameDev *__restrict__ Fs, const uint8_t *__restrict__ req, int *__restrict__ out)
{
    __shared__ MBLocal L;
    const FrameDev F = Fs[0];
    const int lane = LANE();
    const uint8_t *r = req + 1024 * (size_t)blockIdx.x;
    int *o = out + 32 * blockIdx.x;
    for (int i = lane; i < 96; i += 64) { ((uint32_t *)L.fenc)[i] = ((const uint32_t *)r)[i]; ((uint32_t *)L.pred)[i] = ((const uint32_t *)(r + 384))[i]; }
    for (int i = lane; i < 84; i += 64) ((uint8_t *)L.ib_top)[i] = r[768 + i];
    if (lane < 48) ((uint8_t *)L.ib_left)[lane] = r[852 + lane];
    const int avail = *(const int *)(r + 900);
    if (lane == 0) { L.mb_x = L.mb_y = L.mb_xy = 0; L.neighbour = (avail & 1 ? NB_LEFT : 0) | (avail & 2 ? NB_TOP | NB_TOPRIGHT : 0) | (avail == 3 ? NB_TOPLEFT : 0); }
    PCAMV_WAVE_SYNC();
    FrameDev F0 = F;
    F0.psy_rd = 0;
    const int ssd0 = prim_ssd_mb(F0, &L);
    prim_fenc_complexity(F, &L);
    const int ssd1 = prim_ssd_mb(F, &L);
    int h4, h8;
    prim_hadamard_ac16(L.pred, lane, &h4, &h8);
    if (lane == 0) { o[0] = ssd0; o[1] = ssd1; o[2] = h4; o[3] = h8; o[4] = L.fenc_satd_sum; o[5] = L.fenc_sa8d_sum; }
    prim_intra16_satd(F, &L, avail);
    if (lane < 4) o[6 + lane] = L.ccost[lane];
    PCAMV_WAVE_SYNC();
    prim_intra8c_satd(F, &L, avail);
    if (lane < 4) o[10 + lane] = L.ccost[lane];
    PCAMV_WAVE_SYNC();
    prim_intra4_init(&L);
    if (lane < 12) L.slots[lane] = lane;
    prim_intra4_costs(F, &L, 0, 12, 0);
    if (lane < 12) o[14 + lane] = L.ccost[lane];       /* (every mode is computed on the borders as given: availability is the caller's business) */
}
#endif

/* ------------------------------------------------------------------ embedding stage */
#define STC_MAXW 256
struct EmbedDev {
    const pcamv_mb_t *mbs; int n_mb;
    uint8_t *cover, *stego, *message; float *rho; int8_t *flip;
    int *hdr;                 /* [0]=n [1]=m [2]=stc_ok [3]=num_flip [4]=sum(width) [6..7]=(double) sum of rho over the trellis */
    unsigned *cols;           /* [2][STC_MAXW] columns of the two sub-matrices (getMatrix allows widths up to 2^(h-2) = 256, embed.h:286);
                               * cols[2 * STC_MAXW] = shorter, cols[2 * STC_MAXW + 1] = longer */
    unsigned *path;           /* n * 32 words */
    int *rnd;                 /* glibc rand state: r[0..30], f, b */
    long long *lcg;           /* STC column LCG state (embed.h:134) */
    float emrate;
    const uint8_t *user_message; int user_message_len;
    int cap;                  /* capacity of the per-carrier arrays */
    int *car_base;            /* [n_mb] index of each macroblock's first carrier (pass 2 finds its flips there) */
    uint8_t *mbflip;          /* [n_mb] 1 = one of the macroblock's carriers is flipped (k_mb_flips, after the backward pass) */
    unsigned *colinfo;        /* per trellis column, what both Viterbi passes need of it in one word: the (shortened)
                               * matrix column as the forward pass uses it [9:0] and as the backward pass does [22:13], cover bit [10], "last column of its message bit" [11], that
                               * message bit [12] */
};

__device__ __forceinline__ int dev_is01(int d) { return d == 0 || d == 1; }

__device__ int dev_glibc_rand(int *st)
{
    int f = st[31], b = st[32];
    unsigned v = (unsigned)st[f] + (unsigned)st[b];
    st[f] = (int)v;
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    st[31] = f; st[32] = b;
    return (int)((v >> 1) & 0x7fffffff);
}
__device__ int dev_stc_matrix(int width, int height, unsigned *cols, long long *lcg)
{
    if (width >= 2 && width <= 20 && height >= 7 && height <= 12) {
        for (int i = 0; i < width; i++) cols[i] = pcamv_stc_mats_dev[(height - 7) * 400 + (width - 1) * 20 + i];
        return 1;
    }
    if ((1 << (height - 2)) < width) return 0;
    unsigned mask = (1u << (height - 2)) - 1, bop = (1u << (height - 1)) + 1;
    long hold = (long)*lcg;
    for (int i = 0; i < width; i++) {
        unsigned r; int j;
        for (j = -1; j < i;) {
            hold = hold * 214013L + 2531011L;
            r = (((unsigned)(hold >> 16) & 0x7fff & mask) << 1) + bop;
            for (j = 0; j < i; j++) if (cols[j] == r) break;
        }
        cols[i] = r;
    }
    *lcg = hold;
    return 1;
}

#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(1024) k_embed_prepare(const EmbedDev *__restrict__ Es)
{
    const EmbedDev E = Es[blockIdx.x];
    __shared__ int s_cnt[1024];
    const int t = threadIdx.x;
    const int chunk = (E.n_mb + 1023) / 1024;
    const int lo = t * chunk, hi = min(E.n_mb, lo + chunk);
    int cnt = 0, slots[16];
    for (int xy = lo; xy < hi; xy++) {
        const pcamv_mb_t *mb = &E.mbs[xy];
        cnt += carrier_slots(mb->i_type, mb->i_partition, mb->i_sub_partition, mb->used, slots);
    }
    s_cnt[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          /* inclusive Hillis-Steele scan */
        int v = t >= off ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    int base = s_cnt[t] - cnt;
    const int n = s_cnt[1023];
    const float mvc_c1 = 2, mvc_c2 = 0.7f;
    for (int xy = lo; xy < hi; xy++) {
        const pcamv_mb_t *mb = &E.mbs[xy];
        int k = carrier_slots(mb->i_type, mb->i_partition, mb->i_sub_partition, mb->used, slots);
        E.car_base[xy] = base;
        if (!k) continue;
        float rho[16];
        for (int i = 0; i < k; i++) {
            E.cover[base + i] = (uint8_t)((mb->mv[slots[i]][0] + mb->mv[slots[i]][1]) & 1);
            rho[i] = (float)mb->inter_stego_cost[slots[i]];
        }
#define MVD(a, b, c) iabs(mb->mv[a][c] - mb->mv[b][c])
        if (mb->i_type == PCAMV_P_8x8) {
            const uint8_t *sp = mb->i_sub_partition;
            int len = 0;
            if (sp[0] == PCAMV_D_L0_8x8 && sp[1] == PCAMV_D_L0_8x8 && sp[2] == PCAMV_D_L0_8x8 && sp[3] == PCAMV_D_L0_8x8) {
                int c = dev_is01(MVD(0, 4, 0)) + dev_is01(MVD(4, 12, 0)) + dev_is01(MVD(12, 8, 0)) + dev_is01(MVD(8, 0, 0)) +
                        dev_is01(MVD(0, 4, 1)) + dev_is01(MVD(4, 12, 1)) + dev_is01(MVD(12, 8, 1)) + dev_is01(MVD(8, 0, 1));
                float fac = __fadd_rn(__fmul_rn(mvc_c2, (float)c), 1.0f);
                for (int j = 0; j < 4; j++) rho[j] = __fmul_rn(rho[j], fac);
            }
            for (int i = 0; i < 4; i++) {
                if (sp[i] == PCAMV_D_L0_8x8) len += 1;
                else if (sp[i] == PCAMV_D_L0_4x8 || sp[i] == PCAMV_D_L0_8x4) {
                    int b = sp[i] == PCAMV_D_L0_4x8 ? 4 * i + 1 : 4 * i + 2;
                    if (MVD(4 * i, b, 0) + MVD(4 * i, b, 1) < 2) { rho[len] = __fmul_rn(rho[len], mvc_c1); rho[len + 1] = __fmul_rn(rho[len + 1], mvc_c1); }
                    len += 2;
                } else {
                    int q = 4 * i;
                    int c = dev_is01(MVD(q, q + 1, 0)) + dev_is01(MVD(q + 1, q + 3, 0)) + dev_is01(MVD(q + 2, q + 3, 0)) + dev_is01(MVD(q, q + 2, 0)) +
                            dev_is01(MVD(q, q + 1, 1)) + dev_is01(MVD(q + 1, q + 3, 1)) + dev_is01(MVD(q + 2, q + 3, 1)) + dev_is01(MVD(q, q + 2, 1));
                    float fac = __fadd_rn(__fmul_rn(mvc_c2, (float)c), 1.0f);
                    for (int j = 0; j < 4; j++) rho[len + j] = __fmul_rn(rho[len + j], fac);
                    len += 4;
                }
            }
        } else if (mb->i_partition != PCAMV_D_16x16) {
            int b = mb->i_partition == PCAMV_D_8x16 ? 4 : 8;
            if (MVD(0, b, 0) + MVD(0, b, 1) < 2) { rho[0] = __fmul_rn(rho[0], mvc_c1); rho[1] = __fmul_rn(rho[1], mvc_c1); }
        }
#undef MVD
        for (int i = 0; i < k; i++) E.rho[base + i] = rho[i];
        base += k;
    }
    for (int i = t; i < E.cap; i += 1024) { E.stego[i] = 0; E.flip[i] = 0; }
    /* ---- message, sub-matrix schedule (embed.h:340-393), per-column constants ----
     * The schedule "take the longer sub-matrix while the columns used so far stay <= (i + 1) * invalpha + 0.5"
     * has the closed form  columns before message bit i = floor(i * invalpha + 0.5)  (each step adds floor or ceil
     * of invalpha, and the rule picks the one that lands on the next floor; tests/test_stc_schedule.py checks the
     * two agree in the same double arithmetic), so message bits are independent and only the message itself (a
     * lagged-Fibonacci generator) and the sum of rho stay serial, one wave each. */
    __shared__ unsigned s_rnd[64];
    __shared__ unsigned s_cols[2 * STC_MAXW];
    __shared__ int s_ok;
    int m = E.emrate > 1.0f ? (int)E.emrate : (int)__fmul_rn(E.emrate, (float)n);
    if (m < 0) m = 0;
    const bool sched = m > 0 && m <= n;
    const double invalpha = sched ? (double)n / m : 0.0;
    const int shorter = (int)floor(invalpha), longer = (int)ceil(invalpha);
#define STC_BEFORE(i) ((i) == 0 ? 0 : (int)floor((i) * invalpha + 0.5))
    const int nproc = sched ? STC_BEFORE(m) : 0;
    if (t == 64) {
        /* (built in LDS: the random-column generator compares every new column with all earlier ones, a serial walk
         * that should not go through global memory) */
        const int ok = sched && dev_stc_matrix(shorter, 10, s_cols, E.lcg) && dev_stc_matrix(longer, 10, s_cols + STC_MAXW, E.lcg);
        if (ok) {
            for (int k = 0; k < shorter; k++) E.cols[k] = s_cols[k];
            for (int k = 0; k < longer; k++) E.cols[STC_MAXW + k] = s_cols[STC_MAXW + k];
            E.cols[2 * STC_MAXW] = shorter; E.cols[2 * STC_MAXW + 1] = longer;
        }
        E.hdr[0] = n; E.hdr[1] = m; E.hdr[3] = 0;
        E.hdr[4] = ok ? nproc : 0; E.hdr[2] = ok ? -1 : 0;          /* -1: schedule valid, Viterbi pending */
        s_ok = ok;
    }
    if (t < 64) {
        if (E.user_message) {
            for (int i = t; i < imin(m, E.cap); i += 64) E.message[i] = i < E.user_message_len ? E.user_message[i] : 0;     /* m > n (> cap) fails in stc_embed like the reference's; never write past the arrays */
        } else {
            /* glibc TYPE_3 rand(): x[k] = x[k-31] + x[k-3], output x[k] >> 1.  Three outputs are independent of each
             * other, so lanes 0..2 make three per round on a 64-entry ring in LDS.  The stored state is a 31-entry
             * ring with the oldest value at st[31] (f): x[-31 + j] = st[(f + j) % 31]. */
            const int f = E.rnd[31];
            if (t < 31) s_rnd[33 + t] = (unsigned)E.rnd[(f + t) % 31];         /* x[-31 + t] at ring position (-31 + t) & 63 */
            PCAMV_WAVE_SYNC();
            for (int k0 = 0; k0 < m; k0 += 3) {
                const int k = k0 + t;
                if (t < 3 && k < m) {
                    const unsigned v = s_rnd[(k - 31) & 63] + s_rnd[(k - 3) & 63];
                    s_rnd[k & 63] = v;
                    if (k < E.cap) E.message[k] = (uint8_t)(v >> 1 & 1);      /* the stream advances by m whatever the capacity */
                }
                PCAMV_WAVE_SYNC();
            }
            if (t < 31) E.rnd[(f + m + t) % 31] = (int)s_rnd[(m - 31 + t) & 63];
            if (t == 0) { E.rnd[31] = (f + m) % 31; E.rnd[32] = (E.rnd[32] + m) % 31; }
        }
    }
    __syncthreads();
    if (!s_ok) return;
    for (int i = t; i < m; i += 1024) {
        const int start = STC_BEFORE(i);
        const int which = (double)(start + longer) <= (i + 1) * invalpha + 0.5, width = which ? longer : shorter;
        /* shortened columns near the end of the message.  The forward pass drops one row after every message bit i with
         * m - i <= 10 (embed.h:462), the backward pass adds one row back per such bit from the end (embed.h:523): the same
         * mask when m >= 10, not for shorter messages -- the reference's own arithmetic, kept (its stego then does not
         * carry the message) */
        const int left = m - i, drops = imax(0, i - imax(0, m - 10));
        const unsigned fmask = 1023u >> drops, bmask = left >= 10 ? 1023u : (1u << left) - 1, msg = E.message[i] ? 4096u : 0u;
        for (int k = 0; k < width; k++) {
            const unsigned col = s_cols[which * STC_MAXW + k];
            E.colinfo[start + k] = (col & fmask) | (E.cover[start + k] ? 1024u : 0u) | (k == width - 1 ? 2048u : 0u) | msg | (col & bmask) << 13;
        }
    }
#undef STC_BEFORE
    if (t >= 960) {         /* the price of flipping everything, summed in column order like embed.h:448 (the Viterbi's
                             * failure test compares against it): one wave, 64 loads at a time, serial adds */
        const int l = t - 960;
        double total = 0;
        for (int b0 = 0; b0 < nproc; b0 += 64) {        /* columns past the end add +0.0: no effect on a sum of non-negatives */
            const double v = b0 + l < nproc ? (double)E.rho[b0 + l] : 0.0;
            const int lo = __double2loint(v), hi = __double2hiint(v);
#pragma unroll
            for (int i = 0; i < 64; i++) total += __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
        }
        if (l == 0) *(double *)(E.hdr + 6) = total;
    }
}
#endif

/* forward Viterbi over the 1024 trellis states: new[s] = min(p[s] + c_stay, p[s^col] + c_flip), path bit set
 * when the flip branch is <= (embed.h:439-467 evaluated per state; ties and infinities behave identically
 * because both formulations add and compare the same two floats).  The trellis columns are a serial chain with
 * one workgroup barrier each, and what the kernel costs is the time of one link of that chain, so:
 *   - 1024 / NS threads x NS states (s = t + NT j): the per-column bookkeeping is paid once per wave, and a
 *     thread's own p[s] stays in registers;
 *   - the column's constants are fetched one column ahead;
 *   - the fold at the end of a message bit (keep the states whose LSB is that bit, embed.h:469-480) is computed
 *     with the bit's last column instead of in a step of its own;
 *   - the sum of all rho the result is tested against (embed.h:448) is made by k_embed_prepare;
 *   - the barrier waits for LDS traffic only -- __syncthreads() would also drain the path-row stores (s_waitcnt
 *     vmcnt(0)), which nothing in this kernel reads back. */
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int NS>
__global__ void __launch_bounds__(1024 / NS) k_stc_forward(const EmbedDev *__restrict__ Es)
{
    constexpr int NT = 1024 / NS, LOG_NT = NS == 1 ? 10 : NS == 2 ? 9 : 8;
    const EmbedDev E = Es[blockIdx.x];
    __shared__ float s_p[2][1024];
    __shared__ float s_rho[2][256];
    __shared__ unsigned s_info[2][256];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (E.hdr[2] != -1) return;
    const int nproc = E.hdr[4];
    typedef __attribute__((address_space(1))) unsigned long long *gp64w;
    gp64w path = (gp64w)E.path;                         /* global_store, not flat: a flat store also counts on lgkmcnt */
    const float inf = __int_as_float(0x7F800000);
    int cur = 0;
    float p[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) { p[j] = t + j == 0 ? 0.0f : inf; s_p[0][t + NT * j] = p[j]; }
    if (t < 256 && t < nproc) { s_rho[0][t] = E.rho[t]; s_info[0][t] = E.colinfo[t]; }
    __syncthreads();
    unsigned info = s_info[0][0];
    float r = s_rho[0][0];
    for (int index = 0; index < nproc; index++) {
        const int c = index & 255, buf = index >> 8 & 1;
        if (c == 0 && t < 256) {                        /* next 256 columns' constants into the other buffer */
            const int j = index + 256 + t;
            if (j < nproc) { s_rho[buf ^ 1][t] = E.rho[j]; s_info[buf ^ 1][t] = E.colinfo[j]; }
        }
        const int nx = index + 1;
        const unsigned info_n = s_info[nx >> 8 & 1][nx & 255];
        const float r_n = s_rho[nx >> 8 & 1][nx & 255];
        const unsigned column = info & 1023u, xlo = (unsigned)t ^ (column & (NT - 1)), chi = column >> LOG_NT;
        const float c1 = info & 1024u ? r : 0.0f, c2 = info & 1024u ? 0.0f : r;
        float nv[NS];
        unsigned long long bal = 0;
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const float stay = __fadd_rn(p[j], c1), flp = __fadd_rn(s_p[cur][xlo + (((unsigned)j ^ chi) << LOG_NT)], c2);
            const bool bit = flp <= stay;
            nv[j] = bit ? flp : stay;
            const unsigned long long b = __ballot(bit);                 /* states NT j + 64 wv ..: 64-bit word (NT / 64) j + wv of the path row */
            bal = lane == j ? b : bal;
        }
        if (lane < NS) path[(size_t)index * 16 + (NT / 64) * lane + wv] = bal;
        if (info & 2048u) {                             /* last column of a message bit: state s continues as 2s + bit */
#pragma unroll
            for (int j = 0; j < NS; j++) {
                if (NT * j >= 512) { nv[j] = inf; continue; }
                const unsigned sj = t + NT * j, t2 = (2u * sj + (info >> 12 & 1)) & 1023u;
                const float stay2 = __fadd_rn(s_p[cur][t2], c1), flp2 = __fadd_rn(s_p[cur][t2 ^ column], c2);
                nv[j] = sj < 512 ? (flp2 <= stay2 ? flp2 : stay2) : inf;
            }
        }
#pragma unroll
        for (int j = 0; j < NS; j++) { p[j] = nv[j]; s_p[cur ^ 1][t + NT * j] = nv[j]; }
        cur ^= 1;
        LDS_BARRIER();
        info = info_n; r = r_n;
    }
    if (t == 0) {
        const double totalprice = p[0], total = *(const double *)(E.hdr + 6);
        E.hdr[2] = (totalprice >= total) ? 0 : -2;     /* -2: forward ok, backward pending */
    }
}

/* backward walk (embed.h:483-520): one wave, 64 trellis columns per round.  Their path rows sit in registers,
 * word w of every row in lane w, so the serial walk is scalar code around one v_readlane per column; the
 * column's constants (colinfo) come from the lane of the same number.  The walk's state is wave-uniform: the
 * compiler keeps it in SGPRs. */
#ifdef PCAMV_MAIN_TU
static __global__ void __launch_bounds__(64) k_stc_backward(const EmbedDev *__restrict__ Es)
{
    const EmbedDev E = Es[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = E.hdr[0];
    int nf = 0, done_upto = 0;             /* elements [0, done_upto) got their stego bit here */
    if (E.hdr[2] == -2) {
        int index = E.hdr[4] - 1;
        done_upto = E.hdr[4];
        unsigned state = 0;
        while (index >= 0) {
            const int base = index >= 63 ? index - 63 : 0, cnt = index - base + 1;
            unsigned row[64];
#pragma unroll
            for (int e = 0; e < 64; e++) row[e] = (e < cnt && lane < 32) ? E.path[(size_t)(base + e) * 32 + lane] : 0u;
            const unsigned info = lane < cnt ? E.colinfo[base + lane] : 0u;
            unsigned long long out = 0;
#pragma unroll
            for (int e = 63; e >= 0; e--) {
                if (e < cnt) {
                    const unsigned inf = __builtin_amdgcn_readlane(info, e);
                    if (inf & 2048u) state = (state << 1) | (inf >> 12 & 1);
                    const unsigned word = __builtin_amdgcn_readlane(row[e], (state >> 5) & 31);
                    if (word >> (state & 31) & 1) { out |= 1ull << e; state ^= inf >> 13 & 1023u; }
                }
            }
            if (lane < cnt) {       /* stego bit and flip map (encoder.c:1848-1855) of this chunk */
                int st = (int)(out >> lane & 1), f = (int)(info >> 10 & 1) ^ st;
                E.stego[base + lane] = (uint8_t)st; E.flip[base + lane] = (int8_t)f; nf += f;
            }
            index = base - 1;
        }
    }
    /* everything not reached by the Viterbi (failure, m == 0, tail) keeps stego = 0: flip = cover */
    for (int i = done_upto + lane; i < n; i += 64) { int f = E.cover[i]; E.flip[i] = (int8_t)f; nf += f; }
    nf = wave_sum_all(nf);
    if (lane == 0) { E.hdr[3] = nf; E.hdr[2] = E.hdr[2] == -2 ? 1 : 0; }
}
/* per macroblock: is any of its carriers flipped?  The second pass asks this with the macroblock's record instead of looking at the
 * carriers' flags after it (one memory round trip less on its path; 7 of 8 macroblocks at half a bit per carrier have none) */
static __global__ void __launch_bounds__(256) k_mb_flips(const EmbedDev *__restrict__ Es)
{
    const EmbedDev &E = Es[blockIdx.y];
    const int xy = (int)(blockIdx.x * 256 + threadIdx.x);
    if (xy >= E.n_mb) return;
    const int n = E.hdr[0], a = E.car_base[xy], b = xy + 1 < E.n_mb ? E.car_base[xy + 1] : n;
    int any = 0;
    for (int i = a; i < b && i < n; i++) any |= E.flip[i] == 1;
    E.mbflip[xy] = (uint8_t)any;
}
#endif
#endif
