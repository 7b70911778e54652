#!/bin/bash
# tools/dbg/pmc_cache_ab.sh <gops> <tag> [lib]: HBM-side traffic, L2 and L1 counters of one library build (PCAMV_GPU_LIB), one counter group per
# pass (TCC has 4 slots); per-kernel sums -> gpurun_out/pmc_<tag>_{fetch,write,l2,l1}.json (the raw CSVs are deleted: too large to return)
g=${1:-256}; tag=$2; lib=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$lib" ] && export PCAMV_GPU_LIB=$PWD/video-steganography-pcamv_amd/$lib
run() { name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/raw_$name -- python3 bench.py --steps 2 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}_$name.log 2>&1 || exit 1
  python3 tools/dbg/pmc_agg.py gpurun_out/raw_$name gpurun_out/pmc_${tag}_$name.json; rm -rf gpurun_out/raw_$name; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
