#!/bin/bash
# after GPU_MAX_HW_QUEUES=8 (package default) + handle-before-communicator in bench.py: the N > 1 path on one rank under RCCL again, the DP variants
# table, the single-process controls
export TMPDIR=/tmp
OUT=gpurun_out/r4_21; rm -rf $OUT; mkdir -p $OUT; L=$OUT/rccl_hw_queues.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 150 python tools/host_time_dp.py 2>&1 | grep "host enqueue" >> $L; }
run HT_BACKEND=none GPU_MAX_HW_QUEUES=4
run HT_BACKEND=none
run HT_BACKEND=nccl GPU_MAX_HW_QUEUES=4 HT_ONLY_PLAIN=1
run HT_BACKEND=nccl
cat $L
UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_fixed.json 2>> $OUT/err.log
UAD_DP_NO_DEFER=1 UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_fixed_joined.json 2>> $OUT/err.log
timeout 100 python bench.py --quick --steps 50 --warmup 10 > $OUT/plain.json 2>> $OUT/err.log
python - "$OUT" <<'PY' | tee -a $L
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); a = d.get('allreduce') or {}
        print(f.split('/')[-1], 'ms_per_step', d['ms_per_step'], 'without_allreduce', a.get('ms_per_step_without_allreduce'), 'exposed', a.get('exposed_comm_ms'), a.get('backend'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
