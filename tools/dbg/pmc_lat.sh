#!/bin/bash
# tools/dbg/pmc_lat.sh <gops> <tag> [lib]: average latencies (VMEM, LDS, SMEM, instruction fetch: rocprofv3's derived counters) and the
# instruction cache's hit rate, one group per pass; per-kernel sums -> gpurun_out/pmc_<tag>_<group>.json
g=${1:-256}; tag=$2; lib=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$lib" ] && export PCAMV_GPU_LIB=$PWD/video-steganography-pcamv_amd/$lib
run() { name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/raw_$name -- python3 bench.py --steps 1 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$name.log; return 1; }
  python3 tools/dbg/pmc_agg.py gpurun_out/raw_$name gpurun_out/pmc_${tag}_$name.json; rm -rf gpurun_out/raw_$name; }
run vmemlat VmemLatency
run ldslat LdsLatency
run ifetchlat InstrFetchLatency
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_INST_REQ
run smemlat SmemLatency
