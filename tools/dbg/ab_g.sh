#!/bin/bash
# tools/dbg/ab_g.sh "<G list>" "label|ENV=val ENV2=val ..." ...: bench.py main loop only (no extras) at each G under each setting; PCAMV_GPU_LIB selects a library build
GL=$1; shift
for cfg in "$@"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  for G in $GL; do
    env $envs python bench.py --gops $G --steps 3 --warmup 1 --g-sweep '' --cpu-frames 0 --host-io-steps 0 --clip-keyints '' --parity-gops 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label G=$G: %.0f ms/step %.3f M MB/s kernel %.0f ms BER %s' % (j['ms_per_step'], j['value'] / 1e6, j['roofline']['avg_launch_ms'], j['extracted_payload_BER']))"
  done
done
