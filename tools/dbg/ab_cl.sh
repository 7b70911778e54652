#!/bin/bash
# tools/dbg/ab_cl.sh VAR=val ... : closed-loop bench under each environment setting ("-" = none); prints step time
for e in "$@"; do
  echo "== env $e"
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  env $envs timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --gops 256 --cpu-frames 0 --closed-loop 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'MB/s', round(d['ms_per_step'],1), 'ms/step BER', d['extracted_payload_BER'], 'flow ms', round(d['roofline']['avg_launch_ms'],2))" || exit 1
done
