#!/bin/bash
# tools/dbg/ab_env.sh "<gops list>" VAR=val ... : bench under each environment setting ("-" = none)
# (BENCH_ARGS=--open-loop for the first pass alone)
gl="$1"; shift
for e in "$@"; do for g in $gl; do
  echo "== env $e gops=$g"
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  env $envs timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --gops $g --cpu-frames 0 --host-io-steps 0 --g-sweep "" --clip-keyints "" --parity-gops 0 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'MB/s', round(d['ms_per_step'],1), 'ms/step BER', d['extracted_payload_BER'], 'flow kernel ms', round(d['roofline']['avg_launch_ms'],2))" || exit 1
done; done
