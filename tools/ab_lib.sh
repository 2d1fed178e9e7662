#!/bin/bash
# A/B library builds in one session: tools/ab_lib.sh lib_a.so lib_b.so ...   (3 interleaved rounds, ms per frame)
for round in 1 2 3; do
  for lib in "$@"; do
    GS_LIB_PATH=$(pwd)/$lib python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); s = d['stages_ms_untimed_pass']
print('$lib', d['ms_per_step'], 'prepare', s.get('gs_map_prepare'), 'finish', s.get('gs_map_finish'))"
  done
done
