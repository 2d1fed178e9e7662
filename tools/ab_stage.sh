#!/bin/bash
# as ab_lib.sh, printing named stages:  tools/ab_stage.sh STAGE[,STAGE...] lib_a.so lib_b.so ...   (GS_AB_WORKLOAD=c2|c3|...)
stage=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    GS_LIB_PATH=$(pwd)/$lib python3 bench.py --no-cpu-baseline --steps 40 --workload ${GS_AB_WORKLOAD:-c3} 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
text = sys.stdin.read()
try:
    d = json.loads(text); s = d['stages_ms_untimed_pass']
    print('$lib', d['ms_per_step'], ' '.join(f'{k} {s.get(k)}' for k in '$stage'.split(',')))
except Exception as e:
    print('$lib', 'bench failed:', open('gpurun_out/ab_err.txt').read()[-300:])"
  done
done
