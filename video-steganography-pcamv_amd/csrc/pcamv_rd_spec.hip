/*
 * pcamv_rd_spec.hip -- the RD instance of the analysis kernel (pcamv_rd.hip) with the speculative raster chain compiled in
 * (pcamv_kernels.hip.h, mbk_search_spec): a macroblock hands the chain on right after its 16x16 search, its other searches, its RD
 * stage and its successor's searches run side by side on different waves, every macroblock verifies the motion it started from
 * against its predecessor's final one before its RD stage.  For few GOPs in flight (the chip waits for the chains): one wave
 * per SIMD, every register, like pcamv_rd_lo.hip.
 */
#define PCAMV_RD_SPEC 1        /* waves per SIMD */
#include "pcamv_rd.hip"
