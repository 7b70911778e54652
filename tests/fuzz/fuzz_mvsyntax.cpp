/* The MV-syntax extractor (csrc/pcamv_mvsyntax.h, host code of the product) under AddressSanitizer + UBSan on the CPU: real slices
 * with random damage (flipped bytes, truncation, noise) in exact-size heap buffers, so that any read past the end, any overflow
 * and any undefined shift is reported.  Built and run by tests/test_mvsyntax.py::test_parsers_under_sanitizers. */
#define PCAMV_HOST_EMU 1
#include <stdio.h>
#include <vector>
#include <random>
#include "pcamv_host_tables.h"
#include "pcamv_mvsyntax.h"
static std::vector<uint8_t> rd(const char *p) { FILE *f = fopen(p, "rb"); std::vector<uint8_t> v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; }
/* usage: fuzz_mvsyntax <iterations> {<slice.bin> <mb_w> <mb_h> <qp> <cabac>}...   (tests/test_mvsyntax.py dumps the golden slices and runs it) */
int main(int argc, char **argv)
{
    struct Case { const char *f; int w, h, qp, cabac; };
    std::vector<Case> S;
    const int iters = atoi(argv[1]);
    for (int i = 2; i + 4 < argc; i += 5) S.push_back({argv[i], atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4])});
    std::mt19937 rng(7);
    long ok = 0, err = 0;
    for (auto &s : S) {
        std::vector<uint8_t> base = rd(s.f);
        std::vector<pcamv_mb_t> out(s.w * s.h);
        int rc = s.cabac ? pcamv_gpu_parse_pslice_cabac(base.data(), base.size(), s.w, s.h, s.qp, out.data()) : pcamv_gpu_parse_pslice_cavlc(base.data(), base.size(), s.w, s.h, out.data());
        if (rc) { printf("%s: the undamaged slice failed: %d\n", s.f, rc); return 1; }
        for (int k = 0; k < iters; k++) {
            std::vector<uint8_t> d = base;
            int nflip = 1 + rng() % 8;
            for (int i = 0; i < nflip; i++) d[rng() % d.size()] ^= (uint8_t)(1 + rng() % 255);
            if (k % 4 == 0) d.resize(1 + rng() % d.size());
            if (k % 7 == 0) for (auto &b : d) b = (uint8_t)rng();
            /* exact-size heap copy so that any read past the end is caught */
            uint8_t *h = (uint8_t *)malloc(d.size()); memcpy(h, d.data(), d.size());
            rc = s.cabac ? pcamv_gpu_parse_pslice_cabac(h, d.size(), s.w, s.h, (int)(rng() % 52), out.data()) : pcamv_gpu_parse_pslice_cavlc(h, d.size(), s.w, s.h, out.data());
            free(h);
            if (rc) err++; else ok++;
        }
    }
    printf("ok %ld err %ld\n", ok, err);
    return 0;
}
