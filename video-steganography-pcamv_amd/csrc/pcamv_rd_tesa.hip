/*
 * pcamv_rd_tesa.hip -- the RD instance of the analysis kernel (pcamv_rd.hip) with --me tesa compiled in (encoder/me.c:525-600 followed by
 * x264_mb_analyse_p_rd): the reference's default level with its most thorough search.  One wave per SIMD, plain chain (the
 * search itself is what takes the time); the RD stage's inputs are fetched after the searches because the survivor list and the
 * context states share LDS (pcamv_mbkernels.h mbk_search).
 */
#define PCAMV_RD_TESA 1
#ifndef PCAMV_RD_TESA_INLINE
#define PCAMV_SEARCH_CALL 1          /* the search of a partition as a function of its own (pcamv_logic.h): inlined at every call site this unit took 8 minutes to compile */
#endif
#include "pcamv_rd.hip"
