#!/bin/bash
# tools/dbg/kstats.sh <tag> [bench args...] : rocprofv3 kernel trace + stats of a short bench run; keeps the per-kernel summary
# (gpurun_out/kstats_<tag>.csv = what profiles/r0x_*_kernel_stats*.csv are) and the bench line, deletes the trace itself
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 "$@" > $R/gpurun_out/prof_$tag.log 2>&1 || exit 1
find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kstats_$tag.csv \;
rm -rf $R/gpurun_out/prof_$tag
head -14 $R/gpurun_out/kstats_$tag.csv
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-400
