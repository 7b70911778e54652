/*
 * pcamv_rd_spec2.hip -- the speculative raster chain (pcamv_rd_spec.hip) built for 2 waves per SIMD: more chains than the
 * one-wave-per-SIMD build has waves for (each chain keeps ~3 waves busy); see pcamv_gpu.hip for which batch gets which build.
 */
#define PCAMV_RD_SPEC 2
#include "pcamv_rd.hip"
