#!/bin/bash
# rocprofv3 kernel stats of a short bench run; prints the per-kernel table.  usage: tools/prof_bench.sh [workload]
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$$
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $REPO/bench.py --workload ${1:-c3} --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
cd $REPO
python3 tools/kstats.py $OUT 40
