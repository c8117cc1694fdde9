#!/bin/bash
# On the GPU box: rocprofv3 kernel stats + the two PMC traffic passes of the default bench, into gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_profiler.json 2>$OUT/stats.err || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/fetch.err || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/write.err || true
cd $REPO
python tools/traffic.py $OUT/fetch $OUT/write $OUT/traffic_bf16x3.json ${2:-unknown} > $OUT/traffic_table.md
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# keep the merge small: drop the raw per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
