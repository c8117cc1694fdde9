#!/bin/bash
# rocprofv3 kernel stats of the configs[3] bench (parity mode bf16x3) -> gpurun_out/r4_4
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r4_4; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --arch fAnoGAN --variant resnet --math bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_profiler.json 2> $OUT/stats.err || true
cd $REPO
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
head -30 $OUT/kernel_stats.csv | cut -c1-200
