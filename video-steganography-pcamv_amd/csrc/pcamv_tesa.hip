/*
 * pcamv_tesa.hip -- the instance of the analysis kernel that can run --me tesa (encoder/me.c:525-600).
 *
 * The search functions of pcamv_logic.h are templates on TESA: compiled into the common instance, the Hadamard
 * exhaustive search and its run-time choice of the full-pel metric cost every other method 11 %.  The TESA = 1
 * instance is therefore a kernel of its own (dataflow schedule only: PCAMV_SCHED=diag does not take --me tesa), in a
 * translation unit of its own so that the two halves of the library compile side by side (pcamv_amd.build_library:
 * this instance alone takes longer to compile than everything else together).  Nothing else is defined here: the header's other
 * kernels are unused static templates / unreferenced in this unit.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#define PCAMV_TESA_TU 1
#include "pcamv_kernels.hip.h"

void pcamv_launch_flow_tesa(unsigned waves, hipStream_t st, const FrameDev *dF, const FlowDev &fl)
{
    hipLaunchKernelGGL(k_analyse_flow_tesa, dim3(waves), dim3(64), 0, st, dF, fl);
}
