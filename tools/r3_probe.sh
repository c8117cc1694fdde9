#!/bin/bash
# round-3 probe (second session): dispatch timeline of one train step on the final build (rocprofv3 --kernel-trace, no counters)
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/r3/tl; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 20 --warmup 5 --quick --rounds 1 > $OUT/bench.json 2>$OUT/err.log
cd $REPO
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $F adam_kernel > gpurun_out/r3/timeline.txt 2>&1
cat gpurun_out/r3/timeline.txt | cut -c1-110
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
