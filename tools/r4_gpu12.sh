#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) of the k3 kernels of the configs[3] bench -> gpurun_out/r4_12/traffic_fanogan.json
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r4_12; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python $REPO/bench.py --arch fAnoGAN --variant resnet --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/$C.err || true
done
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'convk' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0], r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
    return agg
fe, wr = load(out + '/FETCH_SIZE', 'FETCH_SIZE'), load(out + '/WRITE_SIZE', 'WRITE_SIZE')
rows = []
for k in fe:
    rows.append({'kernel': k[0], 'grid_threads': k[1], 'launches': len(fe[k]), 'fetch_bytes': sum(fe[k]) / len(fe[k]) * 1024 * 2,
                 'write_bytes': (sum(wr[k]) / len(wr[k]) * 1024) if k in wr else None})
rows.sort(key=lambda r: -(r['fetch_bytes'] + (r['write_bytes'] or 0)) * r['launches'])
json.dump({'_note': 'per launch, averaged over the launches of a (kernel, grid); FETCH_SIZE x 1024 x 2 (gfx950: 128-B requests counted at 64 B), WRITE_SIZE x 1024; '
                    'separate --pmc passes with --kernel-trace only', 'rows': rows}, open(out + '/traffic_fanogan.json', 'w'), indent=1)
for r in rows[:14]: print(r)
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
