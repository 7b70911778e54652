#!/bin/bash
# tools/dbg/pmc.sh <tag> <gops>: two PMC passes of one bench step; summaries under gpurun_out/pmc_<tag>_{a,b}
tag=$1; g=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench.py --steps 1 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}_a.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench.py --steps 1 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}_b.log 2>&1 || exit 1
# pass c: lane (EXEC) utilisation of the vector ALU = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU x 4 cycles-per-issue is folded in by the counter)
timeout -k 10 500 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_${tag}_c -- python3 bench.py --steps 1 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}_c.log 2>&1 || exit 1
