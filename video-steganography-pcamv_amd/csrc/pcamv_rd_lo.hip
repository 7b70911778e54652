/*
 * pcamv_rd_lo.hip -- the RD instance of the analysis kernel (pcamv_rd.hip) built for one wave per SIMD: every register a wave
 * can have, nothing spilled; the library launches it while the batch's chains fit the SIMDs anyway (see pcamv_rd.hip).
 */
#define PCAMV_RD_LO 1
#include "pcamv_rd.hip"
