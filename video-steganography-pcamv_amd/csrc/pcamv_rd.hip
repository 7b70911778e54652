/*
 * pcamv_rd.hip -- the instance of the analysis kernel with the RD mode decision of --subme 6 / 7 compiled in
 * (x264_mb_analyse_p_rd / x264_rd_cost_mb, encoder/analyse.c:2117-2186, encoder/rdo.c:139-171: intra SATD thresholds, psy-RD,
 * size-only CABAC / CAVLC, context adaptation; pcamv_logic.h "RD mode decision", pcamv_prims_rd_gpu.h).
 *
 * A kernel of its own for the same reason as the --me tesa instance (pcamv_tesa.hip): compiled into the common instance its
 * code costs the search of --subme <= 5 registers (141 spilled VGPRs, 552 bytes of scratch per lane when it was), and in a
 * translation unit of its own so that the library's instances compile side by side.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#define PCAMV_RD_TU 1
#include "pcamv_kernels.hip.h"

void pcamv_launch_flow_rd(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl)
{
    hipLaunchKernelGGL(k_analyse_flow_rd, dim3(waves), dim3(64), 0, st, dF, fl);
}
int pcamv_flow_rd_waves_per_cu(void)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_analyse_flow_rd, 64, 0) != hipSuccess) return -1;
    return per_cu;
}
#ifdef PCAMV_PROF
/* the phase timers are per translation unit (static __device__): this instance's */
int pcamv_rd_prof_fetch(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcamv_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pcamv_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
