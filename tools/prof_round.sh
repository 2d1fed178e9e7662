#!/bin/bash
# Round-end evidence: kernel stats of the default bench run + PMC passes (each counter group in its own run,
# kernel trace only).  usage (on the GPU box): tools/prof_round.sh TAG   -> gpurun_out/TAG/
set -e
TAG=${1:-r1}
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" && cat $OUT/bench.json | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$name -o run -- python3 $REPO/tools/run_stage.py c3 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name done"
done
cd $REPO
python3 tools/kstats.py $OUT/stats 40 > $OUT/kernel_table.txt
for n in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do python3 tools/pmc_summary.py $OUT/pmc_$n; done > $OUT/pmc_summary.txt
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
