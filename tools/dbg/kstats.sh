#!/bin/bash
# tools/dbg/kstats.sh <tag> [bench args...] : rocprofv3 kernel trace + stats of a short bench run, CSV summary printed
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-io-steps 0 --g-sweep "" "$@" > $R/gpurun_out/prof_$tag.log 2>&1 || exit 1
find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | xargs head -14
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-400
