/*
 * pcamv_gpu.hip -- host side of libpcamv_gpu.so: the C ABI of include/pcamv_gpu.h over the
 * gfx950 kernels of pcamv_kernels.hip.h.  There is no CPU path: every entry point needs a HIP
 * device and fails with PCAMV_ENODEV / PCAMV_EHIP otherwise.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "pcamv_kernels.hip.h"
#include "pcamv_host_tables.h"
#include "pcamv_mvsyntax.h"

#define PCAMV_ABI_VERSION 3
/* chains in a batch up to which the speculative raster schedule is used, and up to which its 1 / 2 waves-per-SIMD builds (measured:
 * pcamv_gpu_batch_create) */
#define PCAMV_SPEC_MAX_CHAINS 3584
#define PCAMV_SPEC1_MAX_CHAINS 320
#define PCAMV_SPEC2_MAX_CHAINS 704
#define NEV 32
#define NRING 8

struct pcamv_ctx;
/* A batch = the set of independent closed-GOP contexts whose frames advance together: every kernel
 * launch carries the same dependency step of all of them (descriptor arrays in device memory). */
struct pcamv_batch {
    int n, device, n_diag, slots_per_mb, max_diag;
    int W, H;
    pcamv_ctx **ctx;
    FrameDev *h_F, *d_F;        /* NRING slots of n descriptors (pinned host / device) */
    EmbedDev *h_E, *d_E;
    hipEvent_t slot_done[NRING];
    int slot_used[NRING], head;
    hipEvent_t ev0[NEV], ev1[NEV];
    int ev_n, ev_head;
    double t_search_ms; int t_search_launches;
    /* dataflow schedule (k_analyse_flow): queue + dependency counters, one persistent launch per step */
    int sched_flow, flow_waves, flow2_waves, closed_loop, rd_lo, rd_spec, stc_ns;
    int b_mbrd, b_tesa;         /* instance of the analysis kernel the batch's contexts need (fixed at creation) */
    unsigned *d_flow;
    FlowDev fl, fl2;          /* queue descriptors of the analysis and of the second pass */
    char err[256];
};

struct pcamv_ctx {
    pcamv_params_t p;
    int device;
    hipStream_t stream;
    pcamv_batch *self;          /* batch of one, used by the per-context entry points */
    pcamv_batch *last;          /* batch that ran this context's most recent analysis */
    FrameDev F;
    EmbedDev E;
    /* device allocations */
    uint8_t *d_fenc[3], *d_raw[3], *d_luma, *d_luma_raster, *d_chroma[2], *d_rec[3];
    int8_t *d_mb_type, *d_ref8, *d_prev_ref, *d_ref8_b;
    int16_t *d_mv, *d_mvr, *d_prev_mv, *d_mvp_aux, *d_mv_b;
    int last_field, prev_internal;   /* ping-pong of the motion field for device-resident chains: which of d_mv (0) / d_mv_b (1) the last analysis wrote */
    pcamv_batch *member_of[16]; int n_member;    /* batches this context belongs to (its own included): told when it closes */
    pcamv_mb_t *d_rec_mb;
    int16_t *d_cost_mv[52];
    uint8_t *d_cover, *d_stego, *d_message, *d_user_msg; unsigned *d_colinfo;
    float *d_rho; int8_t *d_flip; int *d_hdr, *d_rnd; unsigned *d_cols, *d_path; long long *d_lcg;
    int cap;
    int *d_trace;
    uint16_t *d_nnz; int *d_car_base; int8_t *d_flip_user;     /* pass 2 */
    uint8_t *d_mbflip;         /* [n_mb] per macroblock: a carrier of it is flipped in d_flip */
    int rec_pristine;          /* d_rec / d_nnz hold the first pass' reconstruction of the frame last analysed (no second pass has run over it) */
    /* --subme >= 6 */
    uint8_t *d_nb_nz, *d_cabac, *d_cabac_init[52]; int16_t *d_nb_cbp, *d_nb_mvd; uint32_t *d_cabac_tab, *d_dbg_hash;
    char err[256];
};

static int fail(pcamv_ctx *c, int code, const char *fmt, ...)
{
    if (c) { va_list ap; va_start(ap, fmt); vsnprintf(c->err, sizeof(c->err), fmt, ap); va_end(ap); }
    return code;
}
static int bfail(pcamv_batch *b, int code, const char *fmt, ...)
{
    if (b) { va_list ap; va_start(ap, fmt); vsnprintf(b->err, sizeof(b->err), fmt, ap); va_end(ap); }
    return code;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, PCAMV_EHIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define HIPCHKB(b, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bfail(b, PCAMV_EHIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

extern "C" int pcamv_gpu_abi_version(void) { return PCAMV_ABI_VERSION; }
#ifdef PCAMV_PROF
int pcamv_rd_prof_fetch(unsigned long long *out, int reset);
int pcamv_rd_prof_fetch_lo(unsigned long long *out, int reset);
int pcamv_rd_prof_fetch_spec(unsigned long long *out, int reset);
int pcamv_rd_prof_fetch_spec2(unsigned long long *out, int reset);
int pcamv_rd_prof_fetch_tesa(unsigned long long *out, int reset);
int pcamv_rd_prof_fetch_spec4(unsigned long long *out, int reset);
extern "C" int pcamv_gpu_prof_fetch(unsigned long long *out, int reset)
{
    unsigned long long rd[PCAMV_PROF_N];
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcamv_prof), sizeof(unsigned long long) * PCAMV_PROF_N) != hipSuccess) return -1;
    if (reset) { unsigned long long z[PCAMV_PROF_N] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pcamv_prof), z, sizeof(z)) != hipSuccess) return -1; }
    if (pcamv_rd_prof_fetch(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    if (pcamv_rd_prof_fetch_lo(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    if (pcamv_rd_prof_fetch_spec(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    if (pcamv_rd_prof_fetch_spec2(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    if (pcamv_rd_prof_fetch_spec4(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    if (pcamv_rd_prof_fetch_tesa(rd, reset)) return -1;
    for (int i = 0; i < PCAMV_PROF_N; i++) out[i] += rd[i];
    return 0;
}
#endif
extern "C" const char *pcamv_gpu_last_error(const pcamv_ctx_t *c) { return c ? c->err : "no context"; }

template <class T> static hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }

/* glibc srand(seed) state (random_r TYPE_3): the reference draws message bits from rand() with the
 * default seed 1 (encoder.c:1838-1840) */
static void glibc_srand_state(int *st, unsigned seed)
{
    if (!seed) seed = 1;
    st[0] = (int)seed;
    for (int i = 1; i < 31; i++) {
        long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773, w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        st[i] = (int)w;
    }
    int f = 3, b = 0;
    for (int i = 0; i < 310; i++) {
        unsigned v = (unsigned)st[f] + (unsigned)st[b];
        st[f] = (int)v;
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
    }
    st[31] = f; st[32] = b;
}

/* ------------------------------------------------------------------ batches */
extern "C" void pcamv_gpu_batch_destroy(pcamv_batch_t *b)
{
    if (!b) return;
    hipSetDevice(b->device);
    hipDeviceSynchronize();
    for (int i = 0; b->ctx && i < b->n; i++) {          /* contexts closed before the batch have taken themselves out (NULL) */
        pcamv_ctx *c = b->ctx[i];
        if (!c) continue;
        if (c->last == b) c->last = NULL;
        for (int k = 0; k < c->n_member; k++) if (c->member_of[k] == b) { c->member_of[k] = c->member_of[--c->n_member]; break; }
    }
    if (b->h_F) hipHostFree(b->h_F);
    if (b->h_E) hipHostFree(b->h_E);
    hipFree(b->d_F); hipFree(b->d_E);
    if (b->d_flow) hipFree(b->d_flow);
    for (int i = 0; i < NRING; i++) if (b->slot_done[i]) hipEventDestroy(b->slot_done[i]);
    for (int i = 0; i < NEV; i++) { if (b->ev0[i]) hipEventDestroy(b->ev0[i]); if (b->ev1[i]) hipEventDestroy(b->ev1[i]); }
    free(b->ctx);
    delete b;
}

extern "C" int pcamv_gpu_batch_create(pcamv_ctx_t *const *ctxs, int n, pcamv_batch_t **out)
{
    if (!ctxs || n <= 0 || !out) return PCAMV_EINVAL;
    *out = NULL;
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device || ctxs[i]->F.w != ctxs[0]->F.w || ctxs[i]->F.h != ctxs[0]->F.h ||
            ctxs[i]->p.inter != ctxs[0]->p.inter || ctxs[i]->F.b_mbrd != ctxs[0]->F.b_mbrd || ctxs[i]->F.b_cabac != ctxs[0]->F.b_cabac) return PCAMV_EINVAL;
    pcamv_batch *b = new (std::nothrow) pcamv_batch();
    if (!b) return PCAMV_ENOMEM;
    memset((void *)b, 0, sizeof(*b));
    b->n = n; b->device = ctxs[0]->device; b->W = ctxs[0]->F.w; b->H = ctxs[0]->F.h;
    b->b_mbrd = ctxs[0]->F.b_mbrd;
    for (int i = 0; i < n; i++) b->b_tesa |= ctxs[i]->F.me_method == PCAMV_ME_TESA;
    b->ctx = (pcamv_ctx **)malloc(sizeof(pcamv_ctx *) * n);
    for (int i = 0; i < n; i++) b->ctx[i] = ctxs[i];
    const FrameDev &F = ctxs[0]->F;
    b->n_diag = F.mb_w + 2 * (F.mb_h - 1);
    b->max_diag = (F.mb_w + 1) / 2 < F.mb_h ? (F.mb_w + 1) / 2 : F.mb_h;
    b->slots_per_mb = (ctxs[0]->p.inter & PCAMV_ANALYSE_PSUB8x8) ? 16 : 2;
    hipError_t e = hipSetDevice(b->device);
    if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_F, sizeof(FrameDev) * n * NRING, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_E, sizeof(EmbedDev) * n * NRING, hipHostMallocDefault);
    if (e == hipSuccess) e = dalloc(&b->d_F, (size_t)n * NRING);
    if (e == hipSuccess) e = dalloc(&b->d_E, (size_t)n * NRING);
    for (int i = 0; i < NRING && e == hipSuccess; i++) e = hipEventCreateWithFlags(&b->slot_done[i], hipEventDisableTiming);
    for (int i = 0; i < NEV && e == hipSuccess; i++) { e = hipEventCreate(&b->ev0[i]); if (e == hipSuccess) e = hipEventCreate(&b->ev1[i]); }
    /* schedule: PCAMV_SCHED=diag keeps one launch per anti-diagonal (+ separate RCA / encode launches);
     * the default is the dataflow kernel.  Both are the same per-macroblock code. */
    const char *sched = getenv("PCAMV_SCHED");
    b->sched_flow = !(sched && !strcmp(sched, "diag")) && F.n_mb <= 65535 && n <= 65535;
    { const char *v = getenv("PCAMV_STC_STATES"); b->stc_ns = v && (atoi(v) == 2 || atoi(v) == 4) ? atoi(v) : (n >= 1024 ? 4 : 2); }      /* trellis states per thread of the forward Viterbi */
    if (e == hipSuccess && b->sched_flow) {
        const size_t total = (size_t)n * F.n_mb;
        e = dalloc(&b->d_flow, FLOW_CTR_WORDS + 2 * total + (size_t)FLOW_RDONE_STRIDE * n);
        b->fl.ctr = b->d_flow; b->fl.queue = b->d_flow + FLOW_CTR_WORDS; b->fl.dep = (int *)(b->d_flow + FLOW_CTR_WORDS + total);
        b->fl.rdone = b->d_flow + FLOW_CTR_WORDS + 2 * total; b->fl.spec = 0;
        const char *aff = getenv("PCAMV_FLOW_AFFINITY");
        b->fl.nq = (aff && !strcmp(aff, "0")) || n < 8 ? 1 : 8;
        unsigned qb = 0;
        for (int q = 0; q < 8; q++) {
            unsigned ng = q < b->fl.nq ? ((unsigned)n + (unsigned)(b->fl.nq - 1 - q)) / (unsigned)b->fl.nq : 0u;
            b->fl.qbase[q] = qb; b->fl.qcount[q] = ng * (unsigned)F.n_mb; qb += b->fl.qcount[q];
        }
        b->fl.total = (unsigned)total; b->fl.spin_limit = 4u << 20;
        b->fl.n_gop = n; b->fl.n_mb = F.n_mb; b->fl.mb_w = F.mb_w; b->fl.mb_h = F.mb_h; b->fl.fused = 1; b->fl.unit = 1;
        /* one chain per frame: the context states (CABAC), or -- sub-8x8 partitions priced by x264_rd_cost_part -- the non-zero counts
         * / MV differences the macroblock coded before this one leaves in the cache (PCAMV_CHAIN_NZ) */
        b->fl.raster = F.b_mbrd && (F.b_cabac || (ctxs[0]->p.inter & PCAMV_ANALYSE_PSUB8x8));
        int per_cu = 0, n_cu = 0;
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_analyse_flow, 64, 0);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, b->device);
        if (e == hipSuccess && F.b_mbrd) {
            /* which build of the RD instance (pcamv_rd.hip): one wave per SIMD while the chains fit that anyway */
            const char *inst = getenv("PCAMV_RD_INSTANCE");
            /* measured (1080p umh subme 7, MB/s lo / hi): 256 chains 2.31 / 2.21 M, 512: 4.43 / 4.19 M, 1024: 6.84 / 7.77 M -- with one
             * wave per SIMD the lo build has no free wave left at 1024 chains to take the RCA steps off the chains */
            /* Five builds (pcamv_rd*.hip).  With CABAC (raster chains) the ones that hand a chain on speculatively after the 16x16
             * search -- ~3 waves work on a chain then --, at 1, 2 or 4 waves per SIMD by the number of chains; measured (1080p umh
             * subme 7, M MB/s, plain / spec1 / spec2 / spec4): 256 chains 2.51 / 4.60 / - / -, 512: 4.81 / 6.77 / 8.17 / 7.90,
             * 1024: 8.40 / 7.31 / 11.9 / 14.0, 2048: 14.3 / 7.30 / 12.5 / 18.3, 3072: 18.4 / - / - / 19.0, 4096: 19.5 / - / - / 19.3.
             * Without a chain (CAVLC: wavefront order) and for thousands of chains the plain builds: "lo" (1 wave per SIMD) while the
             * chains fit that anyway, else "hi" (4).  PCAMV_RD_INSTANCE=lo|hi|spec|spec2|spec4 and PCAMV_FLOW_SPEC=0|1 override.
             * The speculative chain needs pictures >= FLOW_SPEC_MIN_MBW macroblocks wide. */
            const char *sp = getenv("PCAMV_FLOW_SPEC");
            const int can_spec = b->fl.raster && F.mb_w >= FLOW_SPEC_MIN_MBW;
            int want_spec = inst ? !strncmp(inst, "spec", 4) : (sp ? atoi(sp) != 0 : n <= PCAMV_SPEC_MAX_CHAINS);
            b->rd_spec = can_spec && want_spec ? (inst && !strcmp(inst, "spec") ? 1 : inst && !strcmp(inst, "spec2") ? 2 : inst && !strcmp(inst, "spec4") ? 4 :
                                                   n <= PCAMV_SPEC1_MAX_CHAINS ? 1 : n <= PCAMV_SPEC2_MAX_CHAINS ? 2 : 4) : 0;
            b->rd_lo = !b->rd_spec && (inst && strncmp(inst, "spec", 4) ? !strcmp(inst, "lo") : (b->fl.raster && n <= 2 * n_cu));
            /* sub-8x8 partitions at this level (x264_rd_cost_part): compiled into the two one-wave-per-SIMD builds only */
            if (ctxs[0]->p.inter & PCAMV_ANALYSE_PSUB8x8) { if (b->rd_spec) b->rd_spec = 1; else b->rd_lo = 1; }
            /* --me tesa: its own build (pcamv_rd_tesa.hip), plain chain */
            if (b->b_tesa) { b->rd_spec = 0; b->rd_lo = 0; }
            b->fl.spec = b->rd_spec != 0;
            per_cu = b->b_tesa ? pcamv_flow_rd_waves_per_cu_tesa() : b->rd_spec == 1 ? pcamv_flow_rd_waves_per_cu_spec() : b->rd_spec == 2 ? pcamv_flow_rd_waves_per_cu_spec2() :
                     b->rd_spec == 4 ? pcamv_flow_rd_waves_per_cu_spec4() : b->rd_lo ? pcamv_flow_rd_waves_per_cu_lo() : pcamv_flow_rd_waves_per_cu();
            if (per_cu < 0) e = hipErrorUnknown;
        }
        const char *wv = getenv("PCAMV_FLOW_WAVES");
        long waves = wv ? atol(wv) : (long)per_cu * n_cu;
        if (waves < 1) waves = 1;
        if ((size_t)waves > total) waves = (long)total;
        b->flow_waves = (int)waves;
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pass2_deblock_flow, 64, 0);
        waves = wv ? atol(wv) : (long)per_cu * n_cu;
        if (waves < 1) waves = 1;
        if ((size_t)waves > total) waves = (long)total;
        /* the second pass can take `unit` macroblocks of a row per task (PCAMV_PASS2_UNIT; the dependency graph is the
         * same on the coarser grid).  Measured at G=256: 115.1 / 114.7 / 116.9 / 122.6 ms per step for 1 / 2 / 4 / 8 --
         * its queue traffic is not what bounds it any more.  With thousands of GOPs in flight it is again: 4096 GOPs 2300 / 2277 /
         * 2264 / 2253 ms per step.  Round 3: a task's macroblocks are one LDS tile (P2Unit: one memory round trip, whole cache lines), which is
         * what the run is for now -- 1080p, ms per step for 1 / 8: 1 GOP 316 / 319, 256 GOPs 354 / 353, 512 GOPs 391 / 382, 4096 GOPs: the
         * kernel alone 106 (macroblock by macroblock) -> 65.  Default: 8 from 256 GOPs on, else 1.  Same buffers: the two kernels never overlap. */
        { const char *u = getenv("PCAMV_PASS2_UNIT"); const int unit = u && atoi(u) >= 1 && atoi(u) <= 8 ? atoi(u) : (n >= 256 ? 8 : 1);
          b->fl2 = b->fl; b->fl2.unit = unit; b->fl2.raster = 0; b->fl2.spec = 0; b->fl2.mb_w = (F.mb_w + unit - 1) / unit; b->fl2.n_mb = b->fl2.mb_w * F.mb_h;
          b->fl2.total = (unsigned)n * (unsigned)b->fl2.n_mb;
          unsigned qb2 = 0;
          for (int q = 0; q < 8; q++) { b->fl2.qbase[q] = qb2; b->fl2.qcount[q] = b->fl.qcount[q] / (unsigned)F.n_mb * (unsigned)b->fl2.n_mb; qb2 += b->fl2.qcount[q]; }
          b->fl2.dep = (int *)(b->d_flow + FLOW_CTR_WORDS + b->fl2.total); }
        if ((size_t)waves > b->fl2.total) waves = (long)b->fl2.total;
        b->flow2_waves = (int)waves;
        if (e == hipSuccess) e = hipMemset(b->d_flow, 0, FLOW_CTR_WORDS * sizeof(unsigned));
    }
    if (e != hipSuccess) { pcamv_gpu_batch_destroy(b); return PCAMV_EHIP; }
    for (int i = 0; i < n; i++) if (ctxs[i]->n_member >= 16) { pcamv_gpu_batch_destroy(b); return PCAMV_EINVAL; }
    for (int i = 0; i < n; i++) ctxs[i]->member_of[ctxs[i]->n_member++] = b;
    for (int i = 0; i < n; i++)         /* the kernel instance with --me tesa compiled in exists for the dataflow schedule only */
        if ((ctxs[i]->F.me_method == PCAMV_ME_TESA || ctxs[i]->F.b_mbrd) && !b->sched_flow) { pcamv_gpu_batch_destroy(b); return PCAMV_EUNSUP; }     /* ... and so does the RD mode decision */
    *out = b;
    return 0;
}
extern "C" const char *pcamv_gpu_batch_last_error(const pcamv_batch_t *b) { return b ? b->err : "no batch"; }
extern "C" int pcamv_gpu_batch_set_closed_loop(pcamv_batch_t *b, int on) { if (!b) return PCAMV_EINVAL; b->closed_loop = on != 0; return 0; }
static int flow_check(pcamv_batch *b);
extern "C" int pcamv_gpu_fetch_recon(pcamv_ctx_t *c, uint8_t *const planes[3])
{
    if (!c || !planes) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (c->last) { int rc = flow_check(c->last); if (rc) return fail(c, rc, "%s", c->last->err); }
    const size_t ysz = (size_t)c->F.w * c->F.h;
    for (int i = 0; i < 3; i++)
        if (planes[i]) HIPCHK(c, hipMemcpy(planes[i], c->d_rec[i], i ? ysz / 4 : ysz, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int pcamv_gpu_recon_device(pcamv_ctx_t *c, void *planes[3])
{
    if (!c || !planes) return PCAMV_EINVAL;
    for (int i = 0; i < 3; i++) planes[i] = c->d_rec[i];
    return 0;
}
static const char *dominant_kernel(const pcamv_batch *b)
{
    if (!b || !b->sched_flow) return "k_search_diag";
    if (b->b_mbrd) return "k_analyse_flow_rd";       /* (what batch_create saw: contexts may have been closed since, their slots are NULL) */
    if (b->b_tesa) return "k_analyse_flow_tesa";
    return "k_analyse_flow";
}
extern "C" const char *pcamv_gpu_batch_dominant_kernel(const pcamv_batch_t *b) { return dominant_kernel(b); }
extern "C" int pcamv_gpu_batch_copy_results_async(pcamv_batch_t *b, void *dst_mb, size_t mb_stride, void *dst_flip, size_t flip_stride, void *stream)
{
    if (!b || !dst_mb) return PCAMV_EINVAL;
    HIPCHKB(b, hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < b->n; i++) if (!b->ctx[i]) return bfail(b, PCAMV_EINVAL, "context %d of the batch was closed", i);
    for (int i = 0; i < b->n; i++) {
        pcamv_ctx *c = b->ctx[i];
        const size_t nb = (size_t)c->F.n_mb * sizeof(pcamv_mb_t);
        if (mb_stride < nb || (dst_flip && flip_stride < (size_t)c->cap)) return PCAMV_EINVAL;
        HIPCHKB(b, hipMemcpyAsync((char *)dst_mb + (size_t)i * mb_stride, c->d_rec_mb, nb, hipMemcpyDefault, st));
        if (dst_flip) HIPCHKB(b, hipMemcpyAsync((char *)dst_flip + (size_t)i * flip_stride, c->d_flip, (size_t)c->cap, hipMemcpyDefault, st));
    }
    return 0;
}

/* ------------------------------------------------------------------ contexts */
extern "C" void pcamv_gpu_close(pcamv_ctx_t *c);

static int open_impl(pcamv_ctx *c, const pcamv_params_t *p, int device);
extern "C" int pcamv_gpu_open(const pcamv_params_t *p, int device, pcamv_ctx_t **out)
{
    if (!p || !out) return PCAMV_EINVAL;
    *out = NULL;
    if (p->i_width <= 0 || p->i_height <= 0 || p->i_width % 16 || p->i_height % 16) return PCAMV_EINVAL;
    if (p->i_subpel_refine < 1 || p->i_subpel_refine > 7) return PCAMV_EUNSUP;   /* 8, 9: RD refinement of the MVs (disabled in the fork's P frames anyway, analyse.c:3112) */
    if (p->i_me_method < PCAMV_ME_DIA || p->i_me_method > PCAMV_ME_TESA) return PCAMV_EUNSUP;
    if (p->i_me_method == PCAMV_ME_TESA && p->i_me_range > TESA_MAX_RANGE) return PCAMV_EUNSUP;      /* the survivor list lives in LDS: 32 x 33 positions */
    if (p->i_me_range < 4 || p->i_me_range > 64 || p->i_mv_range < 32) return PCAMV_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PCAMV_ENODEV;
    pcamv_ctx *c = new (std::nothrow) pcamv_ctx();
    if (!c) return PCAMV_ENOMEM;
    memset((void *)c, 0, sizeof(*c));
    c->p = *p; c->device = device; c->last_field = 1;
    const int rc = open_impl(c, p, device);
    if (rc) { pcamv_gpu_close(c); return rc; }          /* one cleanup path: whatever was allocated so far is released */
    *out = c;
    return 0;
}
static int open_impl(pcamv_ctx *c, const pcamv_params_t *p, int device)
{
    HIPCHK(c, hipSetDevice(device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    FrameDev &F = c->F;
    pcamv_frame_set_params(&F, p);
    const size_t ysz = (size_t)F.w * F.h, lsz = (size_t)F.plane_size, csz = (size_t)F.cstride * F.clines;
    for (int i = 0; i < 3; i++) {
        HIPCHK(c, dalloc(&c->d_fenc[i], i ? ysz / 4 : ysz));
        HIPCHK(c, dalloc(&c->d_raw[i], i ? ysz / 4 : ysz));
        HIPCHK(c, dalloc(&c->d_rec[i], i ? ysz / 4 : ysz));
    }
    HIPCHK(c, dalloc(&c->d_luma, 4 * lsz + 64));
    HIPCHK(c, hipMemset(c->d_luma, 0, 4 * lsz + 64));      /* the repeated columns of each plane's last strip are never written */
    if (F.me_method == PCAMV_ME_ESA || F.me_method == PCAMV_ME_TESA) HIPCHK(c, dalloc(&c->d_luma_raster, (size_t)F.stride * F.lines + 64));
    HIPCHK(c, dalloc(&c->d_chroma[0], 2 * (csz + 64))); c->d_chroma[1] = c->d_chroma[0] + csz + 64;   /* one allocation: 32-bit offsets reach both */
    F.cplane_size = (long long)(csz + 64);
    HIPCHK(c, dalloc(&c->d_mb_type, (size_t)F.n_mb)); HIPCHK(c, dalloc(&c->d_ref8, (size_t)F.n_mb * 4)); HIPCHK(c, dalloc(&c->d_prev_ref, (size_t)F.n_mb * 4));
    HIPCHK(c, dalloc(&c->d_mv, (size_t)F.n_mb * 32)); HIPCHK(c, dalloc(&c->d_prev_mv, (size_t)F.n_mb * 32));
    HIPCHK(c, dalloc(&c->d_mv_b, (size_t)F.n_mb * 32)); HIPCHK(c, dalloc(&c->d_ref8_b, (size_t)F.n_mb * 4));
    HIPCHK(c, hipMemset(c->d_mv_b, 0, (size_t)F.n_mb * 64)); HIPCHK(c, hipMemset(c->d_ref8_b, 0xff, (size_t)F.n_mb * 4));
    HIPCHK(c, dalloc(&c->d_mvr, (size_t)F.n_mb * 2)); HIPCHK(c, dalloc(&c->d_mvp_aux, (size_t)F.n_mb * 32));
    HIPCHK(c, dalloc(&c->d_rec_mb, (size_t)F.n_mb));
    HIPCHK(c, hipMemset(c->d_rec_mb, 0, (size_t)F.n_mb * sizeof(pcamv_mb_t)));
    HIPCHK(c, hipMemset(c->d_mv, 0, (size_t)F.n_mb * 64)); HIPCHK(c, hipMemset(c->d_mvr, 0, (size_t)F.n_mb * 4));
    HIPCHK(c, hipMemset(c->d_mvp_aux, 0, (size_t)F.n_mb * 64));
    c->cap = 16 * F.n_mb;
    HIPCHK(c, dalloc(&c->d_cover, (size_t)c->cap)); HIPCHK(c, dalloc(&c->d_stego, (size_t)c->cap)); HIPCHK(c, dalloc(&c->d_message, (size_t)c->cap));
    HIPCHK(c, dalloc(&c->d_user_msg, (size_t)c->cap)); HIPCHK(c, dalloc(&c->d_colinfo, (size_t)c->cap));
    HIPCHK(c, dalloc(&c->d_rho, (size_t)c->cap)); HIPCHK(c, dalloc(&c->d_flip, (size_t)c->cap));
    HIPCHK(c, dalloc(&c->d_hdr, 8)); HIPCHK(c, dalloc(&c->d_rnd, 40)); HIPCHK(c, dalloc(&c->d_cols, 2 * STC_MAXW + 8)); HIPCHK(c, dalloc(&c->d_lcg, 1));
    HIPCHK(c, dalloc(&c->d_path, (size_t)c->cap * 32));
    HIPCHK(c, dalloc(&c->d_nnz, (size_t)F.n_mb)); HIPCHK(c, dalloc(&c->d_car_base, (size_t)F.n_mb)); HIPCHK(c, dalloc(&c->d_flip_user, (size_t)c->cap));
    HIPCHK(c, dalloc(&c->d_mbflip, (size_t)F.n_mb)); HIPCHK(c, hipMemset(c->d_mbflip, 1, (size_t)F.n_mb));
    HIPCHK(c, hipMemset(c->d_nnz, 0, (size_t)F.n_mb * 2)); HIPCHK(c, hipMemset(c->d_car_base, 0, (size_t)F.n_mb * 4));
    HIPCHK(c, hipMemset(c->d_hdr, 0, 8 * sizeof(int)));
    if (F.b_mbrd) {
        HIPCHK(c, dalloc(&c->d_nb_nz, (size_t)F.n_mb * 16)); HIPCHK(c, dalloc(&c->d_nb_cbp, (size_t)F.n_mb)); HIPCHK(c, dalloc(&c->d_nb_mvd, (size_t)F.n_mb * 16));
        HIPCHK(c, dalloc(&c->d_cabac, PCAMV_CHAIN_BYTES)); HIPCHK(c, dalloc(&c->d_cabac_tab, 256));
        HIPCHK(c, hipMemset(c->d_nb_nz, 0, (size_t)F.n_mb * 16)); HIPCHK(c, hipMemset(c->d_nb_cbp, 0, (size_t)F.n_mb * 2)); HIPCHK(c, hipMemset(c->d_nb_mvd, 0, (size_t)F.n_mb * 32));
        HIPCHK(c, hipMemset(c->d_cabac, 0, PCAMV_CHAIN_BYTES));
        uint32_t tab[256]; pcamv_build_cabac_tab(tab);
        HIPCHK(c, hipMemcpy(c->d_cabac_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
        F.nb_nz = c->d_nb_nz; F.nb_cbp = c->d_nb_cbp; F.nb_mvd = c->d_nb_mvd; F.cabac = c->d_cabac; F.cabac_tab = c->d_cabac_tab;
    }
    int rnd[40]; memset(rnd, 0, sizeof(rnd)); glibc_srand_state(rnd, 1);
    HIPCHK(c, hipMemcpy(c->d_rnd, rnd, sizeof(rnd), hipMemcpyHostToDevice));
    long long lcg = 1; HIPCHK(c, hipMemcpy(c->d_lcg, &lcg, sizeof(lcg), hipMemcpyHostToDevice));
    for (int i = 0; i < 3; i++) { F.fenc[i] = c->d_fenc[i]; F.rec[i] = c->d_rec[i]; F.raw[i] = c->d_raw[i]; }
    F.luma_base = c->d_luma; F.chroma_base[0] = c->d_chroma[0]; F.chroma_base[1] = c->d_chroma[1];
    F.luma_raster = c->d_luma_raster;
    for (int k = 0; k < 2; k++) F.chroma[k] = c->d_chroma[k] + (size_t)F.cstride * PCAMV_CPAD + PCAMV_CPAD;
    F.mb_type = c->d_mb_type; F.mv = c->d_mv; F.ref8 = c->d_ref8; F.mvr = c->d_mvr;
    F.prev_mv = c->d_prev_mv; F.prev_ref = c->d_prev_ref; F.have_prev = 0;
    F.rec_mb = c->d_rec_mb; F.mvp_aux = c->d_mvp_aux;
    F.nnz = c->d_nnz; F.car_base = c->d_car_base; F.flip = c->d_flip; F.mbflip = c->d_mbflip;
    EmbedDev &E = c->E;
    E.mbs = c->d_rec_mb; E.n_mb = F.n_mb; E.cover = c->d_cover; E.stego = c->d_stego; E.message = c->d_message; E.rho = c->d_rho;
    E.mbflip = c->d_mbflip;
    E.flip = c->d_flip; E.hdr = c->d_hdr; E.cols = c->d_cols; E.path = c->d_path; E.rnd = c->d_rnd;
    E.lcg = c->d_lcg; E.colinfo = c->d_colinfo; E.cap = c->cap; E.car_base = c->d_car_base; E.user_message = NULL; E.user_message_len = 0; E.emrate = 0;
    pcamv_ctx *one[1] = {c};
    return pcamv_gpu_batch_create(one, 1, &c->self);
}

extern "C" void pcamv_gpu_close(pcamv_ctx_t *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->self) pcamv_gpu_batch_destroy(c->self);
    for (int k = 0; k < c->n_member; k++) {             /* batches that outlive this context must not touch it again */
        pcamv_batch *b = c->member_of[k];
        for (int i = 0; i < b->n; i++) if (b->ctx[i] == c) b->ctx[i] = NULL;
    }
    for (int i = 0; i < 3; i++) { hipFree(c->d_fenc[i]); hipFree(c->d_raw[i]); hipFree(c->d_rec[i]); }
    hipFree(c->d_luma); hipFree(c->d_luma_raster); hipFree(c->d_chroma[0]);
    hipFree(c->d_mb_type); hipFree(c->d_ref8); hipFree(c->d_prev_ref); hipFree(c->d_mv); hipFree(c->d_prev_mv); hipFree(c->d_mvr);
    hipFree(c->d_mvp_aux); hipFree(c->d_rec_mb); hipFree(c->d_mv_b); hipFree(c->d_ref8_b);
    for (int q = 0; q < 52; q++) if (c->d_cost_mv[q]) hipFree(c->d_cost_mv[q]);
    hipFree(c->d_cover); hipFree(c->d_stego); hipFree(c->d_message); hipFree(c->d_colinfo); hipFree(c->d_user_msg); hipFree(c->d_rho);
    hipFree(c->d_flip); hipFree(c->d_hdr); hipFree(c->d_rnd); hipFree(c->d_cols); hipFree(c->d_lcg); hipFree(c->d_path);
    if (c->d_trace) hipFree(c->d_trace);
    hipFree(c->d_nnz); hipFree(c->d_car_base); hipFree(c->d_flip_user); hipFree(c->d_mbflip);
    hipFree(c->d_nb_nz); hipFree(c->d_nb_cbp); hipFree(c->d_nb_mvd); hipFree(c->d_cabac); hipFree(c->d_cabac_tab); hipFree(c->d_dbg_hash);
    for (int q = 0; q < 52; q++) if (c->d_cabac_init[q]) hipFree(c->d_cabac_init[q]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static int ensure_qp(pcamv_ctx *c, int qp)
{
    if (qp < 0 || qp > 51) return fail(c, PCAMV_EINVAL, "qp %d out of range", qp);
    if (!c->d_cost_mv[qp]) {
        int16_t *h = (int16_t *)malloc(PCAMV_COST_MV_LEN * sizeof(int16_t));
        if (!h) return fail(c, PCAMV_ENOMEM, "cost table");
        pcamv_build_cost_mv(qp, h);
        hipError_t e = dalloc(&c->d_cost_mv[qp], (size_t)PCAMV_COST_MV_LEN + 1);      /* + 1: prim_mv_cost fetches the dword that holds an entry */
        if (e == hipSuccess) e = hipMemcpy(c->d_cost_mv[qp], h, PCAMV_COST_MV_LEN * sizeof(int16_t), hipMemcpyHostToDevice);
        free(h);
        if (e != hipSuccess) return fail(c, PCAMV_EHIP, "cost table upload: %s", hipGetErrorString(e));
    }
    pcamv_frame_set_qp(&c->F, &c->p, qp);
    c->F.cost_mv = c->d_cost_mv[qp] + PCAMV_COST_MV_CENTRE;
    if (c->F.b_mbrd) {          /* context states at the slice start for this QP (x264_cabac_context_init, encoder.c:1227) */
        if (!c->d_cabac_init[qp]) {
            uint8_t init[PCAMV_CHAIN_BYTES] = {0};       /* states, then nothing left over from an earlier macroblock */
            pcamv_build_cabac_init(qp, init);
            hipError_t e = dalloc(&c->d_cabac_init[qp], (size_t)PCAMV_CHAIN_BYTES);
            if (e == hipSuccess) e = hipMemcpy(c->d_cabac_init[qp], init, PCAMV_CHAIN_BYTES, hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(c, PCAMV_EHIP, "context initialisation upload: %s", hipGetErrorString(e));
        }
        c->F.cabac_init = c->d_cabac_init[qp];
    }
    return 0;
}
/* diagnostics (parity tests): FNV-1a of the 460 CABAC context states after every macroblock of the following analyses */
extern "C" int pcamv_gpu_debug_state_hash(pcamv_ctx_t *c, int enable)
{
    if (!c) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (enable && !c->d_dbg_hash) { HIPCHK(c, dalloc(&c->d_dbg_hash, (size_t)c->F.n_mb)); HIPCHK(c, hipMemset(c->d_dbg_hash, 0, (size_t)c->F.n_mb * 4)); }
    c->F.dbg_hash = enable ? c->d_dbg_hash : NULL;
    return 0;
}
extern "C" int pcamv_gpu_debug_state_hash_fetch(pcamv_ctx_t *c, uint32_t *out)
{
    if (!c || !out || !c->d_dbg_hash) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_dbg_hash, (size_t)c->F.n_mb * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pcamv_gpu_upload_fenc(pcamv_ctx_t *c, const uint8_t *const plane[3], const int stride[3])
{
    if (!c || !plane || !stride) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 3; i++) {
        int w = c->F.w >> !!i, h = c->F.h >> !!i;
        HIPCHK(c, hipMemcpy2D(c->d_fenc[i], w, plane[i], stride[i], w, h, hipMemcpyHostToDevice));   /* caller memory is pageable: blocking copy */
        c->F.fenc[i] = c->d_fenc[i];
    }
    return 0;
}
extern "C" int pcamv_gpu_set_fenc_device(pcamv_ctx_t *c, const void *y, const void *u, const void *v)
{
    if (!c || !y || !u || !v) return PCAMV_EINVAL;
    c->F.fenc[0] = (const uint8_t *)y; c->F.fenc[1] = (const uint8_t *)u; c->F.fenc[2] = (const uint8_t *)v;
    return 0;
}

/* ------------------------------------------------------------------ batched launches */
/* take the next descriptor slot, fill it from the contexts' current FrameDev/EmbedDev and queue its upload */
static int batch_push_descs(pcamv_batch *b, hipStream_t st, const FrameDev **dF, const EmbedDev **dE, int *slot_out)
{
    int slot = b->head;
    b->head = (b->head + 1) % NRING;
    if (b->slot_used[slot]) HIPCHKB(b, hipEventSynchronize(b->slot_done[slot]));    /* descriptors of this slot no longer in flight */
    FrameDev *hF = b->h_F + (size_t)slot * b->n; EmbedDev *hE = b->h_E + (size_t)slot * b->n;
    for (int i = 0; i < b->n; i++) { hF[i] = b->ctx[i]->F; hE[i] = b->ctx[i]->E; }
    HIPCHKB(b, hipMemcpyAsync(b->d_F + (size_t)slot * b->n, hF, sizeof(FrameDev) * b->n, hipMemcpyHostToDevice, st));
    HIPCHKB(b, hipMemcpyAsync(b->d_E + (size_t)slot * b->n, hE, sizeof(EmbedDev) * b->n, hipMemcpyHostToDevice, st));
    *dF = b->d_F + (size_t)slot * b->n; *dE = b->d_E + (size_t)slot * b->n; *slot_out = slot;
    return 0;
}
static int batch_release_slot(pcamv_batch *b, int slot, hipStream_t st)
{
    HIPCHKB(b, hipEventRecord(b->slot_done[slot], st));
    b->slot_used[slot] = 1;
    return 0;
}

/* what: bit0 plane production, bit1 analysis (search+RCA+encode), bit2 embedding */
static int batch_launch(pcamv_batch *b, int what, hipStream_t st, int timed)
{
    HIPCHKB(b, hipSetDevice(b->device));
    for (int i = 0; i < b->n; i++) {
        pcamv_ctx *c = b->ctx[i];
        if (!c) return bfail(b, PCAMV_EINVAL, "context %d of the batch was closed", i);
        if ((what & 2) && c->prev_internal) {      /* this frame writes the field the last analysis did not write, and reads that one */
            const int wr = !c->last_field;
            c->F.mv = wr ? c->d_mv_b : c->d_mv; c->F.ref8 = wr ? c->d_ref8_b : c->d_ref8;
            c->F.prev_mv = wr ? c->d_mv : c->d_mv_b; c->F.prev_ref = wr ? c->d_ref8 : c->d_ref8_b;
        }
        if (what & 2) c->last_field = c->F.mv == c->d_mv_b;
        if (what & 2) c->rec_pristine = 1;         /* the analysis leaves the first pass' reconstruction + non-zero flags in rec / nnz ... */
        c->F.rec_is_pass1 = (what & 8) ? c->rec_pristine : 0;
        if (what & 8) c->rec_pristine = 0;         /* ... until a second pass has filtered the picture in place */
    }
    const FrameDev *dF; const EmbedDev *dE; int slot;
    int rc = batch_push_descs(b, st, &dF, &dE, &slot);
    if (rc) return rc;
    const FrameDev &F = b->ctx[0]->F;
    const unsigned G = (unsigned)b->n;
    if (what & 1) {
        dim3 g((F.stride / 4 + HP_THREADS - 1) / HP_THREADS, (F.lines + HP_ROWS - 1) / HP_ROWS, G);
        hipLaunchKernelGGL(k_hpel, g, dim3(HP_THREADS), 0, st, dF);
        dim3 gc((F.cstride / 4 + 255) / 256, F.clines, 2 * G);
        hipLaunchKernelGGL(k_chroma_pad, gc, dim3(256), 0, st, dF);
    }
    if (what & 2) {
        int ev = -1, tesa = 0;
        for (int i = 0; i < b->n; i++) tesa |= b->ctx[i]->F.me_method == PCAMV_ME_TESA;       /* the kernel instance with --me tesa compiled in */
        if (timed && !b->sched_flow) { ev = b->ev_head; hipEventRecord(b->ev0[ev], st); }
        for (int i = 0; i < b->n; i++) b->ctx[i]->last = b;
        if (b->sched_flow) {
            hipLaunchKernelGGL(k_flow_init, dim3((b->fl.total + 255) / 256), dim3(256), 0, st, b->fl);
            if (timed) { ev = b->ev_head; hipEventRecord(b->ev0[ev], st); }
            if (F.b_mbrd) {
                if (b->b_tesa) pcamv_launch_flow_rd_tesa((unsigned)b->flow_waves, st, dF, b->fl);
                else if (b->rd_spec == 1) pcamv_launch_flow_rd_spec((unsigned)b->flow_waves, st, dF, b->fl);
                else if (b->rd_spec == 2) pcamv_launch_flow_rd_spec2((unsigned)b->flow_waves, st, dF, b->fl);
                else if (b->rd_spec == 4) pcamv_launch_flow_rd_spec4((unsigned)b->flow_waves, st, dF, b->fl);
                else if (b->rd_lo) pcamv_launch_flow_rd_lo((unsigned)b->flow_waves, st, dF, b->fl);
                else pcamv_launch_flow_rd((unsigned)b->flow_waves, st, dF, b->fl);
            }
            else if (tesa) pcamv_launch_flow_tesa((unsigned)b->flow_waves, st, dF, b->fl);
            else hipLaunchKernelGGL(k_analyse_flow, dim3(b->flow_waves), dim3(64), 0, st, dF, b->fl);
            if (timed) { hipEventRecord(b->ev1[ev], st); b->ev_head = (b->ev_head + 1) % NEV; if (b->ev_n < NEV) b->ev_n++; }
        } else {
            for (int d = 0; d < b->n_diag; d++) {
                int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
                int y_hi = d / 2; if (y_hi > F.mb_h - 1) y_hi = F.mb_h - 1;
                int cnt = y_hi - y_lo + 1;
                if (cnt <= 0) continue;
                hipLaunchKernelGGL(k_search_diag<0>, dim3(cnt, G), dim3(64), 0, st, dF, d);
            }
            if (timed) { hipEventRecord(b->ev1[ev], st); b->ev_head = (b->ev_head + 1) % NEV; if (b->ev_n < NEV) b->ev_n++; }
            hipLaunchKernelGGL(k_rca, dim3(F.n_mb * b->slots_per_mb, G), dim3(64), 0, st, dF, b->slots_per_mb);
            hipLaunchKernelGGL(k_encode, dim3(F.n_mb, G), dim3(64), 0, st, dF);
        }
    }
    if (what & 4) {
        hipLaunchKernelGGL(k_embed_prepare, dim3(G), dim3(1024), 0, st, dE);
        /* 2 trellis states per thread: measured 3.25 / 2.89 / 2.90 ms per 1080p frame for 1 / 2 / 4 */
        /* (one frame alone: 2 trellis states per thread is the fastest chain; thousands of frames: 4 states per thread = 4 waves per frame, so
         * that a CU holds eight frames' trellises instead of four and the batch needs half the rounds: 18.6 -> 13.3 ms per 4096-frame step) */
        if (b->stc_ns == 4) hipLaunchKernelGGL(k_stc_forward<4>, dim3(G), dim3(256), 0, st, dE);
        else hipLaunchKernelGGL(k_stc_forward<2>, dim3(G), dim3(512), 0, st, dE);
        hipLaunchKernelGGL(k_stc_backward, dim3(G), dim3(64), 0, st, dE);
        hipLaunchKernelGGL(k_mb_flips, dim3((F.n_mb + 255) / 256, G), dim3(256), 0, st, dE);
    }
    if (what & 8) {      /* pass 2: final MVs -> reconstruction -> loop filter, same dependency as the search */
        if (b->sched_flow) {
            hipLaunchKernelGGL(k_flow_init, dim3((b->fl2.total + 255) / 256), dim3(256), 0, st, b->fl2);
            hipLaunchKernelGGL(k_pass2_deblock_flow, dim3(b->flow2_waves), dim3(64), 0, st, dF, b->fl2);
        } else {
            for (int d = 0; d < b->n_diag; d++) {
                int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
                int y_hi = d / 2; if (y_hi > F.mb_h - 1) y_hi = F.mb_h - 1;
                int cnt = y_hi - y_lo + 1;
                if (cnt <= 0) continue;
                hipLaunchKernelGGL(k_pass2_deblock_diag, dim3(cnt, G), dim3(64), 0, st, dF, d);
            }
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bfail(b, PCAMV_EHIP, "kernel launch: %s", hipGetErrorString(e));
    return batch_release_slot(b, slot, st);
}
/* after a synchronisation: did the dataflow kernel of the last step give up on a bounded spin? */
static int flow_check(pcamv_batch *b)
{
    if (!b || !b->sched_flow) return 0;
    unsigned bad = 0;
    HIPCHKB(b, hipMemcpy(&bad, b->fl.ctr + FLOW_ERR, sizeof(bad), hipMemcpyDeviceToHost));
    if (bad) {
        hipMemset(b->fl.ctr + FLOW_ERR, 0, sizeof(unsigned));
        return bfail(b, PCAMV_EHIP, "dataflow kernel: queue wait timed out (results of the last step are incomplete)");
    }
    return 0;
}
static int ctx_launch(pcamv_ctx *c, int what)
{
    int rc = batch_launch(c->self, what, c->stream, 1);
    if (rc) { snprintf(c->err, sizeof(c->err), "%s", c->self->err); return rc; }
    return 0;
}

extern "C" int pcamv_gpu_set_ref(pcamv_ctx_t *c, const uint8_t *const plane[3], const int stride[3], const int16_t *prev_mv, const int8_t *prev_ref)
{
    if (!c || !plane || !stride) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 3; i++) {
        int w = c->F.w >> !!i, h = c->F.h >> !!i;
        HIPCHK(c, hipMemcpy2D(c->d_raw[i], w, plane[i], stride[i], w, h, hipMemcpyHostToDevice));
        c->F.raw[i] = c->d_raw[i];
    }
    c->F.have_prev = prev_mv != NULL && prev_ref != NULL && c->p.i_tscale != 0;
    c->F.ref_is_inter = prev_mv != NULL && prev_ref != NULL;      /* the reference picture is a P picture: its macroblock types (analyse.c:369) */
    c->prev_internal = 0;
    c->F.mv = c->d_mv; c->F.ref8 = c->d_ref8;
    c->F.prev_mv = c->d_prev_mv; c->F.prev_ref = c->d_prev_ref;
    if (c->F.have_prev) {
        HIPCHK(c, hipMemcpy(c->d_prev_mv, prev_mv, (size_t)c->F.n_mb * 64, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_prev_ref, prev_ref, (size_t)c->F.n_mb * 4, hipMemcpyHostToDevice));
    }
    int rc = ctx_launch(c, 1);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int pcamv_gpu_set_ref_device(pcamv_ctx_t *c, const void *y, const void *u, const void *v, const void *prev_mv, const void *prev_ref)
{
    if (!c || !y || !u || !v) return PCAMV_EINVAL;
    c->F.have_prev = prev_mv != NULL && prev_ref != NULL && c->p.i_tscale != 0;
    c->F.ref_is_inter = prev_mv != NULL && prev_ref != NULL;
    c->prev_internal = prev_mv == PCAMV_PREV_FIELD_INTERNAL;
    if (c->F.have_prev && !c->prev_internal) { c->F.prev_mv = (const int16_t *)prev_mv; c->F.prev_ref = (const int8_t *)prev_ref; c->F.mv = c->d_mv; c->F.ref8 = c->d_ref8; }
    /* the filter itself runs as the first kernels of the next step (plane production is part of the
     * timed path) and reads the caller's planes in place */
    c->F.raw[0] = (const uint8_t *)y; c->F.raw[1] = (const uint8_t *)u; c->F.raw[2] = (const uint8_t *)v;
    return 0;
}

extern "C" int pcamv_gpu_get_ref_planes(pcamv_ctx_t *c, uint8_t *out, int *stride, int *lines)
{
    if (!c || !out) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    /* the device keeps the planes in strips (pcamv_common.h); the caller gets x264's raster planes */
    const size_t psz = (size_t)c->F.plane_size;
    uint8_t *tmp = (uint8_t *)malloc(4 * psz);
    if (!tmp) return fail(c, PCAMV_EHIP, "out of host memory");
    hipError_t e = hipMemcpy(tmp, c->d_luma, 4 * psz, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(tmp); HIPCHK(c, e); }
    for (int k = 0; k < 4; k++)
        for (int y = 0; y < c->F.lines; y++)
            for (int x = 0; x < c->F.stride; x++)
                out[((size_t)k * c->F.lines + y) * c->F.stride + x] = tmp[k * psz + (size_t)y * PCAMV_LROW + x + (size_t)(x / PCAMV_LSW) * c->F.lskip];
    free(tmp);
    if (stride) *stride = c->F.stride;
    if (lines) *lines = c->F.lines;
    return 0;
}

extern "C" int pcamv_gpu_analyse_pframe(pcamv_ctx_t *c, int qp, int embed, pcamv_mb_t *out_mb, uint8_t *const recon[3])
{
    if (!c || !out_mb) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_qp(c, qp);
    if (rc) return rc;
    c->F.embed = embed;
    if ((rc = ctx_launch(c, 2))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = flow_check(c->self))) return fail(c, rc, "%s", c->self->err);
    HIPCHK(c, hipMemcpy(out_mb, c->d_rec_mb, (size_t)c->F.n_mb * sizeof(pcamv_mb_t), hipMemcpyDeviceToHost));
    if (recon)
        for (int i = 0; i < 3; i++)
            if (recon[i]) HIPCHK(c, hipMemcpy(recon[i], c->d_rec[i], ((size_t)c->F.w * c->F.h) >> (i ? 2 : 0), hipMemcpyDeviceToHost));
    return 0;
}

static int fetch_embed(pcamv_ctx *c, pcamv_embed_t *out)
{
    int hdr[8];
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(hdr, c->d_hdr, sizeof(hdr), hipMemcpyDeviceToHost));
    out->n = hdr[0]; out->m = hdr[1]; out->stc_ok = hdr[2]; out->num_flip = hdr[3];
    if (out->n < 0 || out->m < 0 || out->n > c->cap) return fail(c, PCAMV_EHIP, "embed header corrupt");
    const int m_copy = out->m < c->cap ? out->m : c->cap;          /* a bits-per-frame rate above the capacity: stc_embed failed (m > n), the message array holds cap bits */
    if (out->cover && out->n) HIPCHK(c, hipMemcpy(out->cover, c->d_cover, out->n, hipMemcpyDeviceToHost));
    if (out->rho && out->n) HIPCHK(c, hipMemcpy(out->rho, c->d_rho, (size_t)out->n * 4, hipMemcpyDeviceToHost));
    if (out->stego && out->n) HIPCHK(c, hipMemcpy(out->stego, c->d_stego, out->n, hipMemcpyDeviceToHost));
    if (out->flip && out->n) HIPCHK(c, hipMemcpy(out->flip, c->d_flip, out->n, hipMemcpyDeviceToHost));
    if (out->message && m_copy) HIPCHK(c, hipMemcpy(out->message, c->d_message, m_copy, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pcamv_gpu_embed_pframe(pcamv_ctx_t *c, float emrate, const uint8_t *message, int message_len, pcamv_embed_t *out)
{
    if (!c || !out || emrate <= 0) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (message) {
        if (message_len < 0 || message_len > c->cap) return fail(c, PCAMV_EINVAL, "message_len");
        HIPCHK(c, hipMemcpy(c->d_user_msg, message, message_len, hipMemcpyHostToDevice));
        c->E.user_message = c->d_user_msg; c->E.user_message_len = message_len;
    } else { c->E.user_message = NULL; c->E.user_message_len = 0; }
    c->E.emrate = emrate;
    int rc = ctx_launch(c, 4);
    if (rc) return rc;
    return fetch_embed(c, out);
}

/* Pass 2 of the frame last analysed: final MVs (the record with mv_stego where the flip map says so; the flip
 * map of the last embed_pframe when flips == NULL), reconstruction, loop filter.  out_final / recon /
 * deblocked may be NULL.  The deblocked picture stays on the device (the context's reconstruction planes) and
 * the final motion field becomes the one PCAMV_PREV_FIELD_INTERNAL hands to the next frame. */
extern "C" int pcamv_gpu_pass2_pframe(pcamv_ctx_t *c, const uint8_t *flips, int n_flips, pcamv_mb_t *out_final,
                                      uint8_t *const recon[3], uint8_t *const deblocked[3])
{
    if (!c) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (flips) {
        if (n_flips < 0 || n_flips > c->cap) return fail(c, PCAMV_EINVAL, "n_flips");
        HIPCHK(c, hipMemset(c->d_flip_user, 0, (size_t)c->cap));
        if (n_flips) HIPCHK(c, hipMemcpy(c->d_flip_user, flips, n_flips, hipMemcpyHostToDevice));
        c->F.flip = c->d_flip_user; c->F.mbflip = nullptr;          /* a caller's map: the per-macroblock summary belongs to the embedding stage's own */
        /* carrier index of every macroblock from the record (the embedding stage may not have run) */
        pcamv_mb_t *h = (pcamv_mb_t *)malloc((size_t)c->F.n_mb * sizeof(pcamv_mb_t));
        int *base = (int *)malloc((size_t)c->F.n_mb * sizeof(int));
        if (!h || !base) { free(h); free(base); return fail(c, PCAMV_ENOMEM, "pass2"); }
        HIPCHK(c, hipMemcpy(h, c->d_rec_mb, (size_t)c->F.n_mb * sizeof(pcamv_mb_t), hipMemcpyDeviceToHost));
        int k = 0;
        for (int xy = 0; xy < c->F.n_mb; xy++) {
            base[xy] = k;
            if (!h[xy].used) continue;
            if (h[xy].i_type == PCAMV_P_8x8)
                for (int i = 0; i < 4; i++) k += h[xy].i_sub_partition[i] == PCAMV_D_L0_8x8 ? 1 : h[xy].i_sub_partition[i] == PCAMV_D_L0_4x4 ? 4 : 2;
            else k += h[xy].i_partition == PCAMV_D_16x16 ? 1 : 2;
        }
        hipError_t e = hipMemcpy(c->d_car_base, base, (size_t)c->F.n_mb * sizeof(int), hipMemcpyHostToDevice);
        free(h); free(base);
        if (e != hipSuccess) return fail(c, PCAMV_EHIP, "pass2: %s", hipGetErrorString(e));
        if (k > n_flips) return fail(c, PCAMV_EINVAL, "flip map has %d entries, the record has %d carriers", n_flips, k);
    } else { c->F.flip = c->d_flip; c->F.mbflip = c->d_mbflip; }
    const size_t ysz = (size_t)c->F.w * c->F.h;
    /* pass-2 reconstruction first (for callers that want it before the loop filter), then the filter */
    pcamv_batch *b = c->self;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->F.rec_is_pass1 = c->rec_pristine;
    c->rec_pristine = 0;
    {   /* pass 2 only */
        const FrameDev *dF; const EmbedDev *dE; int slot;
        int rc = batch_push_descs(b, c->stream, &dF, &dE, &slot);
        if (rc) return fail(c, rc, "%s", b->err);
        const FrameDev &F = c->F;
        for (int pass = 0; pass < 2; pass++) {
            for (int d = 0; d < b->n_diag; d++) {
                int y_lo = d - (F.mb_w - 1); y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
                int y_hi = d / 2; if (y_hi > F.mb_h - 1) y_hi = F.mb_h - 1;
                int cnt = y_hi - y_lo + 1;
                if (cnt <= 0) continue;
                if (pass == 0) hipLaunchKernelGGL(k_pass2_diag, dim3(cnt, 1), dim3(64), 0, c->stream, dF, d);
                else hipLaunchKernelGGL(k_deblock_diag, dim3(cnt, 1), dim3(64), 0, c->stream, dF, d);
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            uint8_t *const *dst = pass == 0 ? recon : deblocked;
            if (dst)
                for (int i = 0; i < 3; i++)
                    if (dst[i]) HIPCHK(c, hipMemcpy(dst[i], c->d_rec[i], i ? ysz / 4 : ysz, hipMemcpyDeviceToHost));
        }
        rc = batch_release_slot(b, slot, c->stream);
        if (rc) return fail(c, rc, "%s", b->err);
    }
    if (out_final) {
        HIPCHK(c, hipMemcpy(out_final, c->d_rec_mb, (size_t)c->F.n_mb * sizeof(pcamv_mb_t), hipMemcpyDeviceToHost));
        int16_t *mv = (int16_t *)malloc((size_t)c->F.n_mb * 64);
        if (!mv) return fail(c, PCAMV_ENOMEM, "pass2");
        hipError_t e = hipMemcpy(mv, c->F.mv, (size_t)c->F.n_mb * 64, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(mv); return fail(c, PCAMV_EHIP, "pass2: %s", hipGetErrorString(e)); }
        const int s4 = 4 * c->F.mb_w;
        static const int bx[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3}, by[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};
        for (int xy = 0; xy < c->F.n_mb; xy++) {
            const int mx = xy % c->F.mb_w, my = xy / c->F.mb_w;
            for (int i = 0; i < 16; i++) {
                const int16_t *s = mv + 2 * ((4 * my + by[i]) * s4 + 4 * mx + bx[i]);
                out_final[xy].mv[i][0] = s[0]; out_final[xy].mv[i][1] = s[1]; out_final[xy].ref[i] = 0;
            }
        }
        free(mv);
    }
    return 0;
}

/* one step of every context of the batch on resident inputs: plane production + analysis + embedding */
extern "C" int pcamv_gpu_batch_step(pcamv_batch_t *b, int qp, float emrate, void *stream)
{
    if (!b) return PCAMV_EINVAL;
    HIPCHKB(b, hipSetDevice(b->device));
    for (int i = 0; i < b->n; i++) {
        pcamv_ctx *c = b->ctx[i];
        if (!c) return bfail(b, PCAMV_EINVAL, "context %d of the batch was closed", i);
        int rc = ensure_qp(c, qp);
        if (rc) return bfail(b, rc, "%s", c->err);
        if (!c->F.raw[0]) return bfail(b, PCAMV_EINVAL, "context %d has no reference", i);
        c->F.embed = emrate > 0; c->E.emrate = emrate; c->E.user_message = NULL; c->E.user_message_len = 0;
        c->F.flip = c->d_flip; c->F.mbflip = c->d_mbflip;        /* a closed-loop step applies the flip map of its own embedding stage, never a caller's map left by pass2_pframe */
    }
    hipStream_t st = stream ? (hipStream_t)stream : b->ctx[0]->stream;
    return batch_launch(b, (emrate > 0 ? 7 : 3) | (b->closed_loop ? 8 : 0), st, 1);
}
extern "C" int pcamv_gpu_step_device(pcamv_ctx_t *c, int qp, float emrate, void *stream)
{
    if (!c) return PCAMV_EINVAL;
    int rc = pcamv_gpu_batch_step(c->self, qp, emrate, stream ? stream : (void *)c->stream);
    if (rc) snprintf(c->err, sizeof(c->err), "%s", c->self->err);
    return rc;
}
extern "C" int pcamv_gpu_fetch_results(pcamv_ctx_t *c, pcamv_mb_t *out_mb, pcamv_embed_t *out)
{
    if (!c) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (c->last) { int rc = flow_check(c->last); if (rc) return fail(c, rc, "%s", c->last->err); }
    if (out_mb) HIPCHK(c, hipMemcpy(out_mb, c->d_rec_mb, (size_t)c->F.n_mb * sizeof(pcamv_mb_t), hipMemcpyDeviceToHost));
    if (out) return fetch_embed(c, out);
    return 0;
}

static int batch_kernel_time(pcamv_batch *b, double *avg_ms, int *launches, int reset)
{
    HIPCHKB(b, hipSetDevice(b->device));
    HIPCHKB(b, hipDeviceSynchronize());
    { int rc = flow_check(b); if (rc) return rc; }
    for (int i = 0; i < b->ev_n; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev0[i], b->ev1[i]) == hipSuccess) { b->t_search_ms += ms; b->t_search_launches += b->sched_flow ? 1 : b->n_diag; }
    }
    b->ev_n = 0; b->ev_head = 0;
    if (avg_ms) *avg_ms = b->t_search_launches ? b->t_search_ms / b->t_search_launches : 0;
    if (launches) *launches = b->t_search_launches;
    if (reset) { b->t_search_ms = 0; b->t_search_launches = 0; }
    return 0;
}
extern "C" int pcamv_gpu_batch_kernel_time(pcamv_batch_t *b, const char *kernel, double *avg_ms, int *launches, int reset)
{
    if (!b || !kernel || strcmp(kernel, dominant_kernel(b))) return PCAMV_EINVAL;
    return batch_kernel_time(b, avg_ms, launches, reset);
}
extern "C" int pcamv_gpu_kernel_time(pcamv_ctx_t *c, const char *kernel, double *avg_ms, int *launches, int reset)
{
    if (!c || !kernel || strcmp(kernel, dominant_kernel(c->self))) return PCAMV_EINVAL;
    return batch_kernel_time(c->self, avg_ms, launches, reset);
}

/* diagnostics: log every block-cost evaluation made for macroblock mb during the next analyse call
 * into a device buffer; fetch with pcamv_gpu_trace_fetch.  mb < 0 switches tracing off. */
extern "C" int pcamv_gpu_trace_mb(pcamv_ctx_t *c, int mb)
{
    if (!c) return PCAMV_EINVAL;
#ifndef PCAMV_TRACE
    if (mb >= 0) return fail(c, PCAMV_EUNSUP, "tracing is compiled out: rebuild with -DPCAMV_TRACE");
#endif
    HIPCHK(c, hipSetDevice(c->device));
    if (mb < 0) { c->F.trace = NULL; return 0; }
    if (!c->d_trace) HIPCHK(c, dalloc(&c->d_trace, (size_t)1 + 8 * 4000));
    HIPCHK(c, hipMemset(c->d_trace, 0, (1 + 8 * 4000) * sizeof(int)));
    c->F.trace = c->d_trace; c->F.trace_mb = mb;
    return 0;
}
extern "C" int pcamv_gpu_trace_fetch(pcamv_ctx_t *c, int32_t *out /* 1 + 8*4000 */)
{
    if (!c || !out || !c->d_trace) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_trace, (1 + 8 * 4000) * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pcamv_gpu_block_costs(pcamv_ctx_t *c, int qp, int n, const int32_t *req, int32_t *out)
{
    if (!c || !req || !out || n <= 0) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < n; i++) {
        const int32_t *r = req + 8 * i;
        if (r[0] < 0 || r[0] >= c->F.mb_w || r[1] < 0 || r[1] >= c->F.mb_h || r[2] < 0 || r[2] > 6) return fail(c, PCAMV_EINVAL, "request %d", i);
        int px = r[0] * 16 + r[3] + (r[5] >> 2), py = r[1] * 16 + r[4] + (r[6] >> 2);    /* stay inside the 32-pixel padding */
        if (px < -28 || py < -28 || px + 20 > c->F.w + 28 || py + 20 > c->F.h + 28) return fail(c, PCAMV_EINVAL, "request %d leaves the padded plane", i);
    }
    int rc = ensure_qp(c, qp);
    if (rc) return rc;
    int *d_req = NULL, *d_out = NULL; FrameDev *d_F = NULL;
    HIPCHK(c, dalloc(&d_req, (size_t)n * 8)); HIPCHK(c, dalloc(&d_out, (size_t)n * 3)); HIPCHK(c, dalloc(&d_F, 1));
    HIPCHK(c, hipMemcpy(d_req, req, (size_t)n * 8 * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_F, &c->F, sizeof(FrameDev), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_block_costs, dim3(n), dim3(64), 0, c->stream, (const FrameDev *)d_F, (const int *)d_req, d_out);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_out, (size_t)n * 3 * sizeof(int), hipMemcpyDeviceToHost));
    hipFree(d_req); hipFree(d_out); hipFree(d_F);
    return 0;
}

extern "C" int pcamv_gpu_rd_probe(pcamv_ctx_t *c, int qp, int n, const uint8_t *req, int32_t *out)
{
    if (!c || !req || !out || n <= 0) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_qp(c, qp);
    if (rc) return rc;
    uint8_t *d_req = NULL; int *d_out = NULL; FrameDev *d_F = NULL;
    hipError_t e = dalloc(&d_req, (size_t)n * 1024);
    if (e == hipSuccess) e = dalloc(&d_out, (size_t)n * 32);
    if (e == hipSuccess) e = dalloc(&d_F, 1);
    if (e == hipSuccess) e = hipMemcpy(d_req, req, (size_t)n * 1024, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_F, &c->F, sizeof(FrameDev), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rd_probe, dim3(n), dim3(64), 0, c->stream, (const FrameDev *)d_F, (const uint8_t *)d_req, d_out);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 32 * sizeof(int), hipMemcpyDeviceToHost);
    hipFree(d_req); hipFree(d_out); hipFree(d_F);
    if (e != hipSuccess) return fail(c, PCAMV_EHIP, "rd_probe: %s", hipGetErrorString(e));
    return 0;
}

/* pass-2 substitution (analyse.c:3001-3107) on a fetched record: host-side, trivial */
extern "C" int pcamv_gpu_final_mvs(pcamv_ctx_t *c, pcamv_mb_t *mbs)
{
    if (!c || !mbs) return PCAMV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int hdr[8];
    HIPCHK(c, hipMemcpy(hdr, c->d_hdr, sizeof(hdr), hipMemcpyDeviceToHost));
    int n = hdr[0];
    if (n < 0 || n > c->cap) return fail(c, PCAMV_EHIP, "embed header corrupt");
    int8_t *flip = (int8_t *)malloc(n ? n : 1);
    if (!flip) return fail(c, PCAMV_ENOMEM, "flip");
    hipError_t e = hipMemcpy(flip, c->d_flip, n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(flip); return fail(c, PCAMV_EHIP, "flip download"); }
    int k = 0;
    for (int xy = 0; xy < c->F.n_mb; xy++) {
        int slots[16], cs = 0;
        const pcamv_mb_t *mb = &mbs[xy];
        if (mb->used) {
            if (mb->i_type == PCAMV_P_8x8) {
                for (int i = 0; i < 4; i++)
                    switch (mb->i_sub_partition[i]) {
                    case PCAMV_D_L0_8x8: slots[cs++] = i * 4; break;
                    case PCAMV_D_L0_4x8: slots[cs++] = i * 4; slots[cs++] = i * 4 + 1; break;
                    case PCAMV_D_L0_8x4: slots[cs++] = i * 4; slots[cs++] = i * 4 + 2; break;
                    default: for (int j = 0; j < 4; j++) slots[cs++] = i * 4 + j; break;
                    }
            } else if (mb->i_type == PCAMV_P_L0) {
                slots[cs++] = 0;
                if (mb->i_partition == PCAMV_D_8x16) slots[cs++] = 4;
                else if (mb->i_partition == PCAMV_D_16x8) slots[cs++] = 8;
            }
        }
        for (int i = 0; i < cs && k < n; i++, k++)
            if (flip[k] == 1) { mbs[xy].mv[slots[i]][0] = mbs[xy].mv_stego[slots[i]][0]; mbs[xy].mv[slots[i]][1] = mbs[xy].mv_stego[slots[i]][1]; }
    }
    free(flip);
    return 0;
}

/* sub-matrix columns as the embedder gets them (embed.h:141-199): the published tables for widths 2..20, columns drawn from the
 * code's own LCG (embed.h:134-139) outside that range */
static int host_stc_matrix(int width, int height, unsigned *cols, long long *lcg)
{
    if (width >= 2 && width <= 20 && height >= 7 && height <= 12) {
        for (int i = 0; i < width; i++) cols[i] = pcamv_stc_mats[(height - 7) * 400 + (width - 1) * 20 + i];
        return 1;
    }
    if (!lcg || width < 1 || width > STC_MAXW || (1 << (height - 2)) < width) return 0;
    unsigned mask = (1u << (height - 2)) - 1, bop = (1u << (height - 1)) + 1;
    long hold = (long)*lcg;
    for (int i = 0; i < width; i++) {
        unsigned r = 0; int j;
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
/* syndrome-trellis extractor: H*y over GF(2) with stc_embed's sub-matrix schedule (embed.h:340-393).  lcg: state of the
 * column generator before this frame's embedding (in) / after it (out); NULL = only the tabulated widths */
extern "C" int pcamv_gpu_stc_extract_lcg(const uint8_t *stego, int n, int m, int hgt, int64_t *lcg, uint8_t *message)
{
    if (!stego || !message || n <= 0 || m <= 0 || m > n || hgt < 7 || hgt > 12) return PCAMV_EINVAL;
    double invalpha = (double)n / m;
    int shorter = (int)floor(invalpha), longer = (int)ceil(invalpha);
    if (longer > STC_MAXW) return PCAMV_EUNSUP;
    unsigned *cs = (unsigned *)malloc(2 * (size_t)STC_MAXW * sizeof(unsigned)), *cl = cs + STC_MAXW;
    if (!cs) return PCAMV_ENOMEM;
    long long st = lcg ? (long long)*lcg : 0;
    if (!host_stc_matrix(shorter, hgt, cs, lcg ? &st : NULL) || !host_stc_matrix(longer, hgt, cl, lcg ? &st : NULL)) { free(cs); return PCAMV_EUNSUP; }
    if (lcg) *lcg = (int64_t)st;
    memset(message, 0, m);
    int worm = 0, index = 0;
    for (int i = 0; i < m; i++) {
        int lng = worm + longer <= (i + 1) * invalpha + 0.5;
        int width = lng ? longer : shorter;
        const unsigned *cols = lng ? cl : cs;
        worm += width;
        for (int k = 0; k < width && index < n; k++, index++)
            if (stego[index])
                for (int b = 0; b < hgt && i + b < m; b++) message[i + b] ^= (cols[k] >> b) & 1;
    }
    free(cs);
    return 0;
}
extern "C" int pcamv_gpu_stc_extract(const uint8_t *stego, int n, int m, int hgt, uint8_t *message)
{
    return pcamv_gpu_stc_extract_lcg(stego, n, m, hgt, NULL, message);
}
