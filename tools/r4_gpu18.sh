#!/bin/bash
# deferred joins of the segmented backward (uad_backward_deferred + parallel.DataParallelStep): parity of the segments, RCCL world-1 and gloo two-rank
# DP tests, whole-model tests on the rebuilt library, then bench.py's N > 1 path on one rank under RCCL with and without the deferred joins (same box)
export TMPDIR=/tmp
OUT=gpurun_out/r4_18; rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp_nccl.py -m gpu -q -x 2>&1 | tail -6 > $OUT/pytest_a.log; cat $OUT/pytest_a.log
for i in 1 2; do
  UAD_DP_NO_DEFER=1 UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_joined_$i.json 2>> $OUT/nccl1.err
  UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_deferred_$i.json 2>> $OUT/nccl1.err
done
timeout 100 python bench.py --quick --steps 50 --warmup 10 > $OUT/plain.json 2>> $OUT/nccl1.err
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); a = d.get('allreduce') or {}
        print(f.split('/')[-1], 'ms_per_step', d['ms_per_step'], 'without_allreduce', a.get('ms_per_step_without_allreduce'), 'exposed', a.get('exposed_comm_ms'), a.get('backend'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 400 python -m pytest tests/test_gpu_dp_rehearsal.py tests/test_gpu_trainers.py -m gpu -q -x 2>&1 | tail -6 > $OUT/pytest_b.log; cat $OUT/pytest_b.log
