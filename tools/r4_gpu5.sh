#!/bin/bash
# configs[3] evidence: bench line with the live roofline of the dominant k3 kernel + rocprofv3 kernel stats of the same command
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r4_5; rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py --arch fAnoGAN --variant resnet --math bf16x3 --steps 5 --warmup 2 > $OUT/bench_resnet_bf16x3.json 2> $OUT/bench.err
python - <<'PY'
import json
r = json.load(open('gpurun_out/r4_5/bench_resnet_bf16x3.json'))
print(r['ms_per_step'], r['value'], json.dumps(r['roofline'])[:900])
for k in r.get('k3_kernels', [])[:12]: print(k)
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --arch fAnoGAN --variant resnet --math bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_profiler.json 2> $OUT/stats.err || true
cd $REPO
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
head -14 $OUT/kernel_stats.csv | cut -c1-170
