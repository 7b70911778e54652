#!/bin/bash
g=${1:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_cache_a -- python3 bench.py --steps 2 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_cache_a.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d gpurun_out/pmc_cache_b -- python3 bench.py --steps 2 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_cache_b.log 2>&1 || exit 1
