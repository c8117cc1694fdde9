#!/bin/bash
# bench.py's N > 1 path on one rank under RCCL with 8 hardware queues and the communicator created BEFORE the handle (the order host_time_dp.py measured good);
# and host_time_dp.py with the handle created first (the order that made bench.py's all-reduces cost 0.2 ms each in r4_21)
export TMPDIR=/tmp
OUT=gpurun_out/r4_22; rm -rf $OUT; mkdir -p $OUT; L=$OUT/log.txt; : > $L
UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_q8.json 2>> $OUT/err.log
UAD_DP_NO_DEFER=1 UAD_BENCH_REHEARSAL=nccl1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 \
    bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/nccl1_q8_joined.json 2>> $OUT/err.log
python - "$OUT" <<'PY' | tee -a $L
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); a = d.get('allreduce') or {}
        print(f.split('/')[-1], 'ms_per_step', d['ms_per_step'], 'without_allreduce', a.get('ms_per_step_without_allreduce'), 'exposed', a.get('exposed_comm_ms'), a.get('segments'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
echo "== HT_ENGINE_FIRST=1 (8 queues)" >> $L
HT_BACKEND=nccl HT_ENGINE_FIRST=1 timeout 150 python tools/host_time_dp.py 2>&1 | grep "host enqueue" >> $L
cat $L
