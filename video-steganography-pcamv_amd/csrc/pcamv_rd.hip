/*
 * pcamv_rd.hip -- the instance of the analysis kernel with the RD mode decision of --subme 6 / 7 compiled in
 * (x264_mb_analyse_p_rd / x264_rd_cost_mb, encoder/analyse.c:2117-2186, encoder/rdo.c:139-171: intra SATD thresholds, psy-RD,
 * size-only CABAC / CAVLC, context adaptation; pcamv_logic.h "RD mode decision", pcamv_prims_rd_gpu.h).
 *
 * A kernel of its own for the same reason as the --me tesa instance (pcamv_tesa.hip): compiled into the common instance its
 * code costs the search of --subme <= 5 registers (141 spilled VGPRs, 552 bytes of scratch per lane when it was), and in a
 * translation unit of its own so that the library's instances compile side by side.
 *
 * Two builds of it (this file, and pcamv_rd_lo.hip which includes it with PCAMV_RD_LO defined), differing in the register
 * budget only.  With CABAC a frame is ONE chain of macroblocks (the context states), so a batch of G GOPs keeps G waves busy
 * (+ the RCA work they hand off to whoever is free):
 *   - "lo", 1 wave per SIMD, every register (342 VGPRs in use, lane-derived constants hoisted out of the macroblock loop):
 *     the fastest macroblock.  Used while the chains are few (G <= 2 x CUs): G=64 874 ms per 1080p step, 512: 943 ms = 4.43 M MB/s;
 *   - "hi", 4 waves per SIMD at 128 VGPRs, nothing spilled (the lane number is laundered, pcamv_prims_gpu.h LANE(): with the
 *     lane == k flags hoisted it spilled 149 registers and every reload was a memory round trip in front of its use):
 *     G=1024 7.77 M MB/s (lo: 6.84 M -- no free wave left for the RCA steps), G=4096 19.1 M.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#define PCAMV_RD_TU 1
#ifndef PCAMV_NO_RESIDUAL_CALL
#define PCAMV_RESIDUAL_CALL 1      /* pcamv_prims_rd_gpu.h: the CABAC residual walk as a function of its own */
#endif
/* variant mask of the control code (pcamv_logic.h): 2 = RD mode decision, 4 = speculative raster chain, 8 = sub-8x8 partitions priced by
 * x264_rd_cost_part -- only in the two one-wave-per-SIMD builds, which have the registers for it (compiled into the 4-waves-per-SIMD
 * build it cost 22 spilled VGPRs); batches with --partitions p4x4 at --subme >= 6 run on those (pcamv_gpu_batch_create) */
#if defined(PCAMV_RD_SPEC)          /* pcamv_rd_spec*.hip: the speculative raster chain, PCAMV_RD_SPEC = waves per SIMD (1, 2 or 4) */
#if PCAMV_RD_SPEC == 1
#define PCAMV_RD_LO 1
#define RD_NAME(x) x##_spec
#define PCAMV_RD_VARIANT 14
#elif PCAMV_RD_SPEC == 2
#define RD_NAME(x) x##_spec2
#define PCAMV_RD_VARIANT 6
#else
#define RD_NAME(x) x##_spec4
#define PCAMV_RD_VARIANT 6
#endif
#define PCAMV_RD_OCC PCAMV_RD_SPEC
#elif defined(PCAMV_RD_TESA)        /* pcamv_rd_tesa.hip: the RD mode decision after the Hadamard exhaustive search (--me tesa), one wave per SIMD */
#define PCAMV_RD_LO 1
#define PCAMV_RD_OCC 1
#define PCAMV_RD_VARIANT 11
#define RD_NAME(x) x##_tesa
#elif defined(PCAMV_RD_LO)
#define PCAMV_RD_OCC 1
#define PCAMV_RD_VARIANT 10
#define RD_NAME(x) x##_lo
#else
#ifndef PCAMV_RD_OCC
#define PCAMV_RD_OCC 4
#endif
#define RD_NAME(x) x
#endif
#include "pcamv_kernels.hip.h"

void RD_NAME(pcamv_launch_flow_rd)(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl)
{
    hipLaunchKernelGGL(k_analyse_flow_rd, dim3(waves), dim3(64), 0, st, dF, fl);
}
int RD_NAME(pcamv_flow_rd_waves_per_cu)(void)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_analyse_flow_rd, 64, 0) != hipSuccess) return -1;
    return per_cu;
}
#ifdef PCAMV_PROF
/* the phase timers are per translation unit (static __device__): this instance's */
int RD_NAME(pcamv_rd_prof_fetch)(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcamv_prof), sizeof(unsigned long long) * PCAMV_PROF_N) != hipSuccess) return -1;
    if (reset) { unsigned long long z[PCAMV_PROF_N] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pcamv_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
