# kernel trace of one configs[3] iteration, per launch shape:  gpurun -- 'bash tools/r6_trace.sh [filters]'
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/r6_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $REPO/bench.py --arch fAnoGAN --variant resnet --steps 2 --warmup 1 --quick --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python $REPO/tools/trace_shapes.py $F "$@" > $OUT/shapes.txt; head -70 $OUT/shapes.txt
rm -rf $OUT/t
