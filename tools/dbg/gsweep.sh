#!/bin/bash
# tools/dbg/gsweep.sh "<g list>" VAR=val ...: bench.py's low-G sweep (closed loop, config 3) under environment settings, one JSON line each
G=$1; shift
for kv in "none=0" "$@"; do
  env $kv python bench.py --gops 512 --steps 2 --warmup 1 --g-sweep "$G" --cpu-frames 0 --host-io-steps 0 --clip-keyints '' --parity-gops 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv', ' '.join(f\"G={e['gops']}: {e['ms_per_step']:.0f} ms {e['value']/1e6:.3f}M\" for e in j['g_sweep']), 'BER', j['extracted_payload_BER'])"
done
