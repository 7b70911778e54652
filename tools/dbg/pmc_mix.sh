#!/bin/bash
# one --pmc pass (instruction mix) of the bench command with a given library
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "$lib" != "-" ] && export PCAMV_GPU_LIB=$PWD/video-steganography-pcamv_amd/$lib
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/raw_$tag -- python3 bench.py --steps 1 --warmup 1 --gops 4096 --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 > gpurun_out/pmc_${tag}.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}.log; exit 1; }
python3 tools/dbg/pmc_agg.py gpurun_out/raw_$tag gpurun_out/pmc_${tag}.json; rm -rf gpurun_out/raw_$tag
python3 - <<PY
import json
j=json.load(open('gpurun_out/pmc_${tag}.json'))
for k in j:
    if 'analyse_flow_rd' in k:
        print('$tag', {c: round(v['sum']/v['dispatches']/33423360,1) for c,v in j[k].items()})
PY
