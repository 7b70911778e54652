#!/bin/bash
# tools/dbg/pmc_sq.sh <gops> <tag> [lib] [bench args]: the SQ counter groups of tools/dbg/pmc.sh (activity, instruction mix, lane utilisation) and the
# derived latencies, one group per pass; per-kernel sums -> gpurun_out/pmc_<tag>_<group>.json (raw CSVs deleted on the box)
g=${1:-256}; tag=$2; lib=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$lib" ] && [ "$lib" != "-" ] && export PCAMV_GPU_LIB=$PWD/video-steganography-pcamv_amd/$lib
run() { name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/raw_$name -- python3 bench.py --steps 1 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 $BARGS > gpurun_out/pmc_${tag}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$name.log; return 1; }
  python3 tools/dbg/pmc_agg.py gpurun_out/raw_$name gpurun_out/pmc_${tag}_$name.json; rm -rf gpurun_out/raw_$name; }
BARGS="$*"
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
run c SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
