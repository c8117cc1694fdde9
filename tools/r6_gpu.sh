#!/bin/bash
# Round-6 GPU calls (same steps as round 5; new ones at the end), one parameterised script (replaces the per-call r4_gpu<N>.sh files):  gpurun -- 'bash tools/r5_gpu.sh <step>'
# Every step writes under gpurun_out/r5_<step>/ ; what is kept is copied to profiles/r05_<step>_*.
export TMPDIR=/tmp
STEP=${1:?step}
OUT=gpurun_out/r6_$STEP; rm -rf $OUT; mkdir -p $OUT
Q="--quick --no-cpu-baseline"
case $STEP in
a)  # verdict item 1, step A: issue-port micro-benchmark, shader clock under load, phase clocks of the dec3 filter gradient and of dec3.fwd, same-box baseline
    timeout 120 tools/issue_ubench > $OUT/issue_ubench.log 2>&1
    UAD_DBG=160 timeout 200 python bench.py --steps 5 --warmup 2 --rounds 1 $Q > $OUT/w_tr_phase.json 2> $OUT/w_tr_phase.log
    UAD_DBG=64 timeout 200 python bench.py --steps 20 --warmup 30 --rounds 1 $Q > $OUT/d16s_phase.json 2> $OUT/d16s_phase.log
    timeout 200 python bench.py --steps 50 --warmup 10 $Q > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err
    timeout 200 python bench.py --steps 30 --warmup 10 --math f32 $Q > $OUT/bench_f32.json 2> $OUT/bench_f32.err
    cat $OUT/issue_ubench.log; grep -h "w5 CB\|wg0 w0\|wg5 w3\|wg10 w6" $OUT/w_tr_phase.log | head -40; grep -h "d16s wg" $OUT/d16s_phase.log | head -8
    python - <<'PY'
import json
for f in ('bench_bf16x3', 'bench_f32'):
    try:
        d = json.load(open(f'gpurun_out/r5_a/{f}.json'))
        r = d['roofline']
        print(f, d['ms_per_step'], d['value'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('clock_ghz_measured'), r.get('frac_at_measured_clock'))
    except Exception as e:
        print(f, 'failed', e)
PY
    ;;
b)  # f32 pipelined filter gradient (parity: same bits as the round-2 kernel) + A/B; library-issued RCCL tests; DP fault test
    timeout 600 python -m pytest tests/test_gpu_dp_nccl.py tests/test_gpu_dp_rehearsal.py "tests/test_gpu_knobs.py::test_bottleneck_sibling_exchange_is_bounded_and_reports" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest_dp.log
    cat $OUT/pytest_dp.log
    UAD_MATH=f32 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_large.py -m gpu -q -x -p no:cacheprovider -k "conv_w" 2>&1 | tail -5 > $OUT/pytest_w_f32.log
    cat $OUT/pytest_w_f32.log
    timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > $OUT/pytest_model.log
    cat $OUT/pytest_model.log
    for r in 1 2; do
      UAD_NO_W_F32P=1 timeout 200 python bench.py --steps 30 --warmup 5 --math f32 $Q > $OUT/f32_old_$r.json 2>/dev/null
      timeout 200 python bench.py --steps 30 --warmup 5 --math f32 $Q > $OUT/f32_new_$r.json 2>/dev/null
    done
    python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_b/f32_*.json')):
    try:
        d = json.load(open(f)); k = d['kernels']
        print(f.split('/')[-1], d['ms_per_step'], d['value'], ' '.join(f"{t}={k[t]['ms']*1e3:.1f}" for t in k if t.endswith('wgrad')))
    except Exception as e:
        print(f, 'failed', e)
PY
    ;;
c)  # A/B of two library builds: ablibs/libA.so (before) vs the tree's (after); [TESTS="pytest args"] MODES="bf16x3 f32" bash tools/r5_gpu.sh c
    [ -n "$TESTS" ] && { UAD_MATH=bf16x3 timeout 900 python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4; }
    for r in 1 2 3; do for v in A B; do for m in ${MODES:-bf16x3}; do
      L=$PWD/ablibs/libA.so; [ $v = B ] && L=${LIBB:-$PWD/unsupervised_anomaly_detection_brain_mri_amd/libuad_hip.so}
      UAD_LIB=$L timeout 200 python bench.py --steps 40 --warmup 5 --math $m $Q > $OUT/${m}_${v}_$r.json 2>/dev/null
    done; done; done
    python tools/ab_table.py $OUT $TAGS
    ;;
e)  # same-box A/B of two ENVIRONMENTS on the tree's library: ENVA="K=V .." ENVB="K=V .." MODES="bf16x3 f32" [TESTS="pytest args"] bash tools/r5_gpu.sh e
    [ -n "$TESTS" ] && timeout 900 python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
    for r in 1 2 3; do for v in A B; do for m in ${MODES:-bf16x3}; do
      E="$ENVA"; [ $v = B ] && E="$ENVB"
      env $E timeout 200 python bench.py --steps 40 --warmup 5 --math $m $Q > $OUT/${m}_${v}_$r.json 2>/dev/null
    done; done; done
    echo "A: $ENVA | B: $ENVB"
    python tools/ab_table.py $OUT
    ;;
f)  # phase clocks of the filter-gradient kernel of the given math mode:  MODE=f32 bash tools/r5_gpu.sh f
    UAD_DBG=160 timeout 200 python bench.py --steps 5 --warmup 2 --rounds 1 --math ${MODE:-f32} $Q > $OUT/phase.json 2> $OUT/phase.log
    grep -h "w5 CB\|wg0 w\|wg5 w\|wg10 w" $OUT/phase.log | head -60
    ;;
g)  # library-issued RCCL on one rank: step time of the plain step, the torch process-group path and the library path (own stream / side stream), same box
    export GPU_MAX_HW_QUEUES=8
    timeout 300 python tools/host_time_dp.py > $OUT/rccl_own_stream.log 2>&1; cat $OUT/rccl_own_stream.log | grep -v Warning
    UAD_AR_STREAM=side timeout 300 python tools/host_time_dp.py 2>&1 | grep "library RCCL\|plain" > $OUT/rccl_side_stream.log; cat $OUT/rccl_side_stream.log
    UAD_AR_SKIP=1 timeout 300 python tools/host_time_dp.py 2>&1 | grep "library RCCL\|plain" > $OUT/rccl_own_stream_skip.log; echo "UAD_AR_SKIP=1 (everything but the ncclAllReduce call):"; cat $OUT/rccl_own_stream_skip.log
    UAD_AR_SKIP=1 UAD_AR_STREAM=side timeout 300 python tools/host_time_dp.py 2>&1 | grep "library RCCL\|plain" > $OUT/rccl_side_stream_skip.log; cat $OUT/rccl_side_stream_skip.log
    UAD_BENCH_REHEARSAL=nccl1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 50 --warmup 10 $Q > $OUT/bench_nccl1.json 2> $OUT/bench_nccl1.err
    python -c "import json; d = json.load(open('$OUT/bench_nccl1.json')); print(d['ms_per_step'], d['value'], json.dumps(d.get('allreduce'))[:600])"
    ;;
h)  # bench.py's N > 1 path under RCCL on one rank (UAD_BENCH_REHEARSAL=nccl1): torch path, library path, and orderings / queue counts of the library path
    run1() { tag=$1; shift; env "$@" UAD_BENCH_REHEARSAL=nccl1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 50 --warmup 10 $Q > $OUT/$tag.json 2> $OUT/$tag.err
      python -c "import json; s = open('$OUT/$tag.json').read(); d = json.loads(s[s.index('{'):]); a = d['allreduce']; print('$tag', 'stdout clean' if s.lstrip().startswith('{') else 'STDOUT POLLUTED', d['ms_per_step'], 'without all-reduce', a['ms_per_step_without_allreduce'], 'exposed', a['exposed_comm_ms'])"; }
    run1 torch_pg UAD_DP_LIBRARY_AR=0
    run1 library UAD_X=0
    run1 library_own_stream UAD_AR_STREAM=own
    [ -n "$MORE" ] && { run1 library_q16 GPU_MAX_HW_QUEUES=16; run1 library_q4 GPU_MAX_HW_QUEUES=4; }
    ;;
i)  # planner: minimum workgroup count of a spatial launch (small batches: the 8x8 / 16x16 layers of the 16-slice workloads)
    for v in 256 128 64 32; do
      UAD_SPATIAL_MIN_WGS=$v timeout 300 python bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --restore-steps 50 --no-cpu-baseline > $OUT/gmvae_$v.json 2>/dev/null
      UAD_SPATIAL_MIN_WGS=$v timeout 300 python bench.py --arch ceVAE --steps 40 --warmup 5 $Q > $OUT/cevae_$v.json 2>/dev/null
      python -c "
import json
g = json.load(open('$OUT/gmvae_$v.json')); c = json.load(open('$OUT/cevae_$v.json'))
print('min_wgs $v: gmvae restore', g['value'], 'slices/s', g['config']['ms_per_restore_iteration'], 'ms/iter |', ' '.join(f\"{t}={g['kernels'][t]['ms']*1e3:.0f}\" for t in ('enc4.fwd','enc4.dgrad','dec0.fwd','dec0.dgrad','enc3.fwd','dec1.dgrad') if t in g['kernels']), '| cevae16', c['value'], c['ms_per_step'])"
    done
    ;;
j)  # restoration through the pattern word: parity (GMVAE / VAE_You restore tests, small-batch model tests for the planner change), then same-box A/B
    timeout 900 python -m pytest tests/test_gpu_gmvae.py tests/test_gpu_vae_you.py tests/test_gpu_model.py tests/test_gpu_shapes.py tests/test_gpu_cevae.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
    for r in 1 2; do for v in UAD_NO_RESTORE_BITS=1 UAD_X=0; do
      env $v timeout 300 python bench.py --arch GMVAE_spatial --steps 1 --warmup 1 --restore-steps 50 --no-cpu-baseline > $OUT/gmvae_${v%%=*}_$r.json 2>/dev/null
      python -c "
import json
g = json.load(open('$OUT/gmvae_${v%%=*}_$r.json')); k = g['kernels']
print('$v', g['config']['ms_per_restore_iteration'], 'ms/iter ->', round(16 / (150 * g['config']['ms_per_restore_iteration'] * 1e-3), 1), 'slices/s at 150 steps |', ' '.join(f\"{t}={k[t]['ms']*1e3:.0f}\" for t in list(k)[:8]))"
    done; done
    ;;
k)  # configs[3] (ResNet f-AnoGAN): parity tests of the GAN handle, then same-box A/B of two library builds (ablibs/libA.so = before, the tree's = after)
    timeout 900 python -m pytest tests/test_gpu_fanogan.py tests/test_gpu_ops_resnet.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
    for r in 1 2; do for v in A B; do
      L=$PWD/ablibs/libA.so; [ $v = B ] && L=$PWD/unsupervised_anomaly_detection_brain_mri_amd/libuad_hip.so
      UAD_LIB=$L timeout 300 python bench.py --arch fAnoGAN --variant resnet --steps 5 --warmup 2 --no-cpu-baseline > $OUT/resnet_${v}_$r.json 2>/dev/null
      python -c "
import json
d = json.load(open('$OUT/resnet_${v}_$r.json')); k = d.get('k3_kernels') or {}
print('$v', d['ms_per_step'], 'ms/step', d['value'], 'slices/s')"
    done; done
    ;;
bits)  # bit identity of two library builds over three seeded VAE steps (x_hat, every gradient tensor, parameters): MODES="bf16x3 f32" bash tools/r6_gpu.sh bits
    for m in ${MODES:-bf16x3}; do
      UAD_LIB=$PWD/ablibs/libA.so timeout 300 python tools/ab_bits.py $m > $OUT/bits_${m}_A.txt 2> $OUT/bits_${m}_A.err
      timeout 300 python tools/ab_bits.py $m > $OUT/bits_${m}_B.txt 2> $OUT/bits_${m}_B.err
      if diff -q $OUT/bits_${m}_A.txt $OUT/bits_${m}_B.txt > /dev/null && [ -s $OUT/bits_${m}_A.txt ]; then echo "$m: BIT-IDENTICAL ($(wc -l < $OUT/bits_${m}_A.txt) digests)"; else echo "$m: DIFFERENT"; diff $OUT/bits_${m}_A.txt $OUT/bits_${m}_B.txt | head -20; tail -3 $OUT/bits_${m}_B.err; fi
    done
    ;;
sweep)  # one environment variable over a list of values, default bench line each:  VAR=UAD_W5_TARGET VALS="384 448 512 640" [MODES=bf16x3] bash tools/r6_gpu.sh sweep
    for r in 1 2; do for v in $VALS; do for m in ${MODES:-bf16x3}; do
      env $VAR=$v timeout 200 python bench.py --steps 40 --warmup 5 --math $m $Q > $OUT/${m}_${v}_$r.json 2>/dev/null
      python - <<PY
import json
try:
    j = json.load(open('$OUT/${m}_${v}_$r.json')); k = j['kernels']
    w = sum(k[t]['ms'] for t in k if t.endswith('.wgrad') and t not in ('enc0.wgrad', 'bott.wgrad')) * 1e3
    print('$VAR=$v', '$m', 'round $r', j['ms_per_step'], 'ms  sum k5 wgrad %.1f' % w, ' '.join('%s=%.1f' % (t, k[t]['ms'] * 1e3) for t in ('dec3.wgrad', 'dec2.wgrad', 'enc1.wgrad', 'enc3.wgrad', 'dec3.dgrad', 'dec3.fwd', 'enc3.fwd', 'dec0.fwd')))
except Exception as e:
    print('$VAR=$v failed', e)
PY
    done; done; done
    ;;
ln)  # clustered one-pass LayerNorm on the 64 x 64 maps: parity, then same-box comparison against the two-pass kernels (UAD_NO_LNQ=1); the pixel-lane sweep 128 / 64 / 32 of the first run is in profiles/README.md
    timeout 900 python -m pytest tests/test_gpu_fanogan.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
    for r in 1 2; do for v in ${BASE:-UAD_NO_LNQ=1} UAD_X=0 $EXTRA; do
      env $v timeout 300 python bench.py --arch fAnoGAN --variant resnet --steps 5 --warmup 2 --no-cpu-baseline > $OUT/resnet_${v}_$r.json 2>/dev/null
      python -c "
import json
d = json.load(open('$OUT/resnet_${v}_$r.json'))
print('$v', d['ms_per_step'], 'ms/step', d['value'], 'slices/s')"
    done; done
    ;;
*)  echo "unknown step $STEP"; exit 2;;
esac
