#!/bin/bash
# On the GPU box: round 4's closing evidence at HEAD -- full GPU suite, default bench, rocprofv3 kernel stats in BOTH math modes, one step's kernel
# timeline, HBM-traffic PMC passes in both modes, SQ instruction / cycle counters of the default mode, and the configs[2] / [3] / [4] bench lines with
# their kernel stats -- into gpurun_out/final_r04/
COMMIT=${1:-unknown}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/final_r04; rm -rf $OUT; mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu_full.log
fi
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --steps 10 --warmup 3 --quick > $OUT/bench_under_profiler.json 2>$OUT/stats.err || true
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python $REPO/tools/timeline.py {} adam > $OUT/step_timeline.txt 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python $REPO/bench.py --steps 10 --warmup 3 --quick --math f32 > $OUT/bench_under_profiler_f32.json 2>$OUT/stats_f32.err || true
find $OUT/stats_f32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_f32.csv
for M in bf16x3 f32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$M -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 --math $M > /dev/null 2>$OUT/fetch_$M.err || true
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$M -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 --math $M > /dev/null 2>$OUT/write_$M.err || true
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq_insts -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 > /dev/null 2>$OUT/sq_insts.err || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq_cycles -- python $REPO/bench.py --steps 2 --warmup 1 --quick --rounds 1 > /dev/null 2>$OUT/sq_cycles.err || true
# the other BASELINE configs' bench lines + kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_resnet -- python $REPO/bench.py --arch fAnoGAN --variant resnet --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fanogan_resnet64_bench_under_profiler.json 2>$OUT/stats_resnet.err || true
find $OUT/stats_resnet -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/fanogan_resnet64_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cevae -- python $REPO/bench.py --arch ceVAE --steps 10 --warmup 3 --quick > $OUT/cevae_bench_under_profiler.json 2>$OUT/stats_cevae.err || true
find $OUT/stats_cevae -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cevae_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gmvae -- python $REPO/bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --quick --restore-steps 20 > $OUT/gmvae_bench_under_profiler.json 2>$OUT/stats_gmvae.err || true
find $OUT/stats_gmvae -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/gmvae_restore_kernel_stats.csv
cd $REPO
timeout 600 python bench.py --arch fAnoGAN --variant resnet --steps 5 --warmup 2 > $OUT/bench_fanogan_resnet64.json 2> $OUT/bench_fanogan.err
timeout 300 python bench.py --arch ceVAE --no-cpu-baseline > $OUT/bench_cevae_b16.json 2> $OUT/bench_cevae.err
timeout 600 python bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_gmvae_restore.json 2> $OUT/bench_gmvae.err
for M in bf16x3 f32; do
  python tools/traffic.py $OUT/fetch_$M $OUT/write_$M $OUT/traffic_$M.json $COMMIT $M > $OUT/traffic_table_$M.md 2>$OUT/traffic_$M.err
done
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for tag in ('sq_insts', 'sq_cycles'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + '/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:90], r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    with open(out + '/pmc_' + tag + '.csv', 'w') as fo:
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            fo.write(k[0] + ' | grid=' + k[1] + ' | n=' + str(len(next(iter(cs.values())))) + ' | ' + ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(cs.items())) + '\n')
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
rm -rf $OUT/stats $OUT/stats_f32 $OUT/stats_resnet $OUT/stats_cevae $OUT/stats_gmvae $OUT/fetch_* $OUT/write_* $OUT/sq_insts $OUT/sq_cycles 2>/dev/null
ls -la $OUT; tail -3 $OUT/pytest_gpu_full.log 2>/dev/null; head -c 400 $OUT/bench_default.json; echo; head -4 $OUT/pmc_sq_insts.csv
