#!/bin/bash
# HBM-side traffic of the analysis kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC has 4 slots)
g=${1:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_write.log 2>&1 || exit 1
