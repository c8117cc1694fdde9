#!/bin/bash
# exact-fp32 mode with the final 1x1 conv + loss fused into the last ConvT's epilogue (conv5_d_kernel<..., FIN>): same-box A/B against
# UAD_NO_FUSED_FINAL_F32=1, then the FULL GPU suite on this build, the default bench line, and the f32 mode's kernel stats + traffic passes
COMMIT=${1:-unknown}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r4_16; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  UAD_NO_FUSED_FINAL_F32=1 timeout 300 python bench.py --quick --math f32 --steps 50 --warmup 10 --rounds 3 > $OUT/ab_unfused_$i.json 2>>$OUT/ab.err
  timeout 300 python bench.py --quick --math f32 --steps 50 --warmup 10 --rounds 3 > $OUT/ab_fused_$i.json 2>>$OUT/ab.err
done
python - "$OUT" <<'PY' > $OUT/ab_summary.txt 2>&1
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + '/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernels', {})
        pick = {t: round(v['ms'] * 1e3, 1) if isinstance(v, dict) else v for t, v in k.items() if t in ('dec3.fwd', 'final.fwd+bwd', 'loss.finalize', 'dec3.wgrad', 'dec3.dgrad')}
        print(f.split('/')[-1], d['ms_per_step'], d['value'], pick)
    except Exception as e:
        print(f, 'unreadable', e)
PY
cat $OUT/ab_summary.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest_gpu_full.log
tail -4 $OUT/pytest_gpu_full.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 600 $OUT/bench_default.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python $REPO/bench.py --steps 10 --warmup 3 --quick --math f32 > $OUT/bench_under_profiler_f32.json 2>$OUT/stats_f32.err || true
find $OUT/stats_f32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_f32.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_f32 -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 --math f32 > /dev/null 2>$OUT/fetch_f32.err || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_f32 -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 --math f32 > /dev/null 2>$OUT/write_f32.err || true
cd $REPO
python tools/traffic.py $OUT/fetch_f32 $OUT/write_f32 $OUT/traffic_f32.json $COMMIT f32 > $OUT/traffic_table_f32.md 2>$OUT/traffic_f32.err
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
rm -rf $OUT/stats_f32 $OUT/fetch_f32 $OUT/write_f32 2>/dev/null
ls -la $OUT; head -12 $OUT/kernel_stats_f32.csv | cut -c1-160
