#!/bin/bash
# On the GPU box: round 6's closing evidence at HEAD, in the order the bench lines need it -- rocprofv3 kernel stats + HBM-traffic PMC passes of EVERY bench
# command first (all five BASELINE configs: VAE in three math modes, ceVAE, spatial-GMVAE restoration, f-AnoGAN ResNet), written into profiles/ of the box's
# copy so that the bench lines that follow embed them; SQ counters + a step timeline; then the bench lines (default, ceVAE, GMVAE restoration, GMVAE full
# volume, f-AnoGAN ResNet, one-rank RCCL rehearsal), the scoring-kernel line and the full GPU suite -> gpurun_out/final_r06/
#   gpurun --timeout 3400 -- 'bash tools/final_round6.sh <commit>'      (SKIP_TESTS=1: without the suite; ONLY="vae fanogan ..." : a subset of the profile groups)
COMMIT=${1:-unknown}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/final_r06; rm -rf $OUT; mkdir -p $OUT
P=$REPO/profiles
Q="--quick --no-cpu-baseline"
ONLY=${ONLY:-"vae cevae gmvae fanogan sq"}
cd /tmp
prof3() {   # prof3 <tag> <bench args...>: stats + FETCH_SIZE + WRITE_SIZE passes of one bench command (PMC passes with --kernel-trace only)
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- python $REPO/bench.py "$@" > $OUT/${tag}_bench_under_profiler.json 2>$OUT/stats_$tag.err || true
  find $OUT/stats_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${tag}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python $REPO/bench.py "$@" > /dev/null 2>$OUT/fetch_$tag.err || true
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$tag -- python $REPO/bench.py "$@" > /dev/null 2>$OUT/write_$tag.err || true
}
has() { case " $ONLY " in *" $1 "*) return 0;; *) return 1;; esac; }
if has vae; then
  prof3 vae_bf16x3 --steps 10 --warmup 3 $Q
  find $OUT/stats_vae_bf16x3 -name "*kernel_trace.csv" | head -1 | xargs -I{} python $REPO/tools/timeline.py {} adam > $OUT/step_timeline.txt 2>/dev/null || true
  prof3 vae_f32 --steps 10 --warmup 3 $Q --math f32
  prof3 vae_bf16x6 --steps 10 --warmup 3 $Q --math bf16x6
  for M in bf16x3 f32 bf16x6; do
    python $REPO/tools/traffic.py $OUT/fetch_vae_$M $OUT/write_vae_$M $OUT/traffic_$M.json $COMMIT $M > $OUT/traffic_table_$M.md 2>$OUT/traffic_$M.err
    cp $OUT/traffic_$M.json $P/r06_traffic_$M.json
  done
  cp $OUT/vae_bf16x3_kernel_stats.csv $P/r06_z_kernel_stats.csv; cp $OUT/vae_f32_kernel_stats.csv $P/r06_z_kernel_stats_f32.csv; cp $OUT/vae_bf16x6_kernel_stats.csv $P/r06_z_kernel_stats_bf16x6.csv
fi
if has cevae; then
  prof3 cevae_b16 --arch ceVAE --steps 10 --warmup 3 $Q
  python $REPO/tools/evidence.py $OUT/fetch_cevae_b16 $OUT/write_cevae_b16 $OUT/cevae_b16_kernel_stats.csv $OUT/evidence_cevae_b16.json $COMMIT "bench.py --arch ceVAE --steps 10 --warmup 3 --quick" > $OUT/evidence_cevae_b16.txt 2>&1
  cp $OUT/evidence_cevae_b16.json $P/r06_evidence_cevae_b16.json
fi
if has gmvae; then
  prof3 gmvae_restore --arch GMVAE_spatial --steps 1 --warmup 1 --restore-steps 20 $Q
  python $REPO/tools/evidence.py $OUT/fetch_gmvae_restore $OUT/write_gmvae_restore $OUT/gmvae_restore_kernel_stats.csv $OUT/evidence_gmvae_restore_b16.json $COMMIT "bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --restore-steps 20 --quick" > $OUT/evidence_gmvae_restore.txt 2>&1
  cp $OUT/evidence_gmvae_restore_b16.json $P/r06_evidence_gmvae_restore_b16.json
fi
if has fanogan; then
  prof3 fanogan_resnet64 --arch fAnoGAN --variant resnet --steps 2 --warmup 1 $Q
  python $REPO/tools/evidence.py $OUT/fetch_fanogan_resnet64 $OUT/write_fanogan_resnet64 $OUT/fanogan_resnet64_kernel_stats.csv $OUT/evidence_fanogan_resnet64.json $COMMIT "bench.py --arch fAnoGAN --variant resnet --steps 2 --warmup 1 --quick" > $OUT/evidence_fanogan_resnet64.txt 2>&1
  cp $OUT/evidence_fanogan_resnet64.json $P/r06_evidence_fanogan_resnet64.json; cp $OUT/fanogan_resnet64_kernel_stats.csv $P/r06_z_fanogan_resnet64_kernel_stats.csv
fi
if has sq; then
  # ---- SQ counters of the default mode (instruction mix per kernel: VALU per MFMA) and of bf16x6
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq_insts -- python $REPO/bench.py --steps 2 --warmup 1 --rounds 1 $Q > /dev/null 2>$OUT/sq_insts.err || true
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq_cycles -- python $REPO/bench.py --steps 2 --warmup 1 --rounds 1 $Q > /dev/null 2>$OUT/sq_cycles.err || true
  python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for tag in ('sq_insts', 'sq_cycles'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + '/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:110], r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    with open(out + '/pmc_' + tag + '.csv', 'w') as fo:
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            extra = ''
            if 'SQ_INSTS_VALU' in cs and 'SQ_INSTS_MFMA' in cs and sum(cs['SQ_INSTS_MFMA']) > 0:
                extra = ' VALU_per_MFMA=%.2f' % (sum(cs['SQ_INSTS_VALU']) / sum(cs['SQ_INSTS_MFMA']))
            fo.write(k[0] + ' | grid=' + k[1] + ' | n=' + str(len(next(iter(cs.values())))) + ' | ' + ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(cs.items())) + extra + '\n')
PY
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
rm -rf $OUT/stats_* $OUT/fetch_* $OUT/write_* $OUT/sq_insts $OUT/sq_cycles 2>/dev/null
cd $REPO
[ "${NO_BENCH:-0}" = "1" ] && { ls $OUT; head -12 $OUT/pmc_sq_insts.csv 2>/dev/null; head -30 $OUT/fanogan_resnet64_kernel_stats.csv 2>/dev/null | cut -c1-200; exit 0; }
# ---- the bench lines (they read the files written above)
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python bench.py --arch ceVAE --no-cpu-baseline > $OUT/bench_cevae_b16.json 2> $OUT/bench_cevae.err
timeout 600 python bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_gmvae_restore.json 2> $OUT/bench_gmvae.err
timeout 600 python bench.py --arch GMVAE_spatial --volume --steps 3 --warmup 1 > $OUT/bench_gmvae_volume.json 2> $OUT/bench_gmvae_volume.err
timeout 600 python bench.py --arch fAnoGAN --variant resnet --steps 5 --warmup 2 > $OUT/bench_fanogan_resnet64.json 2> $OUT/bench_fanogan.err
GPU_MAX_HW_QUEUES=8 UAD_BENCH_REHEARSAL=nccl1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 100 --warmup 10 --quick --no-cpu-baseline > $OUT/bench_nccl1.json 2> $OUT/bench_nccl1.err
timeout 300 python tools/eval_bench.py > $OUT/eval_bench.txt 2> $OUT/eval_bench.err
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest_gpu_full.log
fi
ls -la $OUT; tail -3 $OUT/pytest_gpu_full.log 2>/dev/null; head -c 300 $OUT/bench_default.json; echo; head -4 $OUT/pmc_sq_insts.csv; tail -2 $OUT/eval_bench.txt
python - <<PY
import json
for f in ('bench_default', 'bench_cevae_b16', 'bench_gmvae_restore', 'bench_gmvae_volume', 'bench_fanogan_resnet64', 'bench_nccl1'):
    try:
        d = json.load(open('$OUT/' + f + '.json')); r = d.get('roofline') or {}
        print(f, d['value'], d['unit'], d['ms_per_step'], '|', r.get('kernel'), r.get('frac'), 'clock', r.get('clock_ghz_measured'), 'traffic', (r.get('traffic') or {}).get('bytes'), 'rocprof', (r.get('rocprof') or {}).get('frac') if isinstance(r.get('rocprof'), dict) else r.get('rocprof'))
        for o in d.get('other_math_modes') or []:
            print('   ', o['math'], o['value'], o['ms_per_step'], o['roofline'].get('kernel'), o['roofline'].get('frac'), 'step / fp32 MFMA peak', o.get('step_fraction_of_fp32_mfma_peak'))
    except Exception as e:
        print(f, 'failed:', e)
PY
