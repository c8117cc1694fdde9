#!/bin/bash
# Usage (on the GPU box): tools/pmc.sh <tag> "<counters>" [bench args]   -> gpurun_out/pmc_<tag>.csv (per-kernel means)
# Counters go in their own pass with --kernel-trace only (never combined with sys/hip traces).
set -e
TAG=$1; shift
CNT=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || true
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
files = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '.csv', 'w') as fo:
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        line = k + ' | n=' + str(len(next(iter(cs.values())))) + ' | ' + ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(cs.items()))
        fo.write(line + '\n')
print(open(out + '.csv').read()[:6000])
PY
