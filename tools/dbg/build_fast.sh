#!/bin/bash
# Development build: compile the library's translation units side by side, keeping the objects under /tmp/pobj so that
# only what changed is recompiled (pass the units to rebuild: gpu tesa rd rd_lo; default all), then link both the normal
# library and the -DPCAMV_PROF one (libpcamv_gpu_prof.so).   tools/dbg/build_fast.sh [units...] [--prof]
set -e
cd "$(dirname "$0")/../../video-steganography-pcamv_amd"
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result"
UNITS=""; PROF=0
for a in "$@"; do if [ "$a" = "--prof" ]; then PROF=1; else UNITS="$UNITS $a"; fi; done
[ -z "$UNITS" ] && UNITS="gpu tesa rd rd_lo rd_spec rd_spec2 rd_spec4 rd_tesa"
mkdir -p /tmp/pobj
FLAGS="$FLAGS $EXTRA"        # e.g. EXTRA=-DPCAMV_RD_OCC=2
PIDS=""
for u in $UNITS; do
  hipcc $FLAGS -c -o /tmp/pobj/$u.o csrc/pcamv_$u.hip & PIDS="$PIDS $!"
  if [ $PROF = 1 ]; then hipcc $FLAGS -DPCAMV_PROF -c -o /tmp/pobj/${u}_prof.o csrc/pcamv_$u.hip & PIDS="$PIDS $!"; fi
done
for p in $PIDS; do wait $p || { echo "COMPILE FAILED"; exit 1; }; done
hipcc --offload-arch=gfx950 -fPIC -shared -o ${OUT:-libpcamv_gpu.so} /tmp/pobj/gpu.o /tmp/pobj/tesa.o /tmp/pobj/rd.o /tmp/pobj/rd_lo.o /tmp/pobj/rd_spec.o /tmp/pobj/rd_spec2.o /tmp/pobj/rd_spec4.o /tmp/pobj/rd_tesa.o
[ -f /tmp/pobj/gpu_prof.o ] && [ -f /tmp/pobj/tesa_prof.o ] && [ -f /tmp/pobj/rd_prof.o ] && [ -f /tmp/pobj/rd_lo_prof.o ] && [ -f /tmp/pobj/rd_spec_prof.o ] && [ -f /tmp/pobj/rd_spec2_prof.o ] && [ -f /tmp/pobj/rd_spec4_prof.o ] && [ -f /tmp/pobj/rd_tesa_prof.o ] && hipcc --offload-arch=gfx950 -fPIC -shared -o libpcamv_gpu_prof.so /tmp/pobj/gpu_prof.o /tmp/pobj/tesa_prof.o /tmp/pobj/rd_prof.o /tmp/pobj/rd_lo_prof.o /tmp/pobj/rd_spec_prof.o /tmp/pobj/rd_spec2_prof.o /tmp/pobj/rd_spec4_prof.o /tmp/pobj/rd_tesa_prof.o
ls -la *.so
