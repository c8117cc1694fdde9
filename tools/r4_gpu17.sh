#!/bin/bash
# bench.py's N > 1 code path under backend nccl (RCCL) on the one rank a one-GPU box has (UAD_BENCH_REHEARSAL=nccl1), launched exactly as the driver launches
# the multi-GPU bench; then smoke() at HEAD
export TMPDIR=/tmp
OUT=gpurun_out/r4_17; rm -rf $OUT; mkdir -p $OUT
UAD_BENCH_REHEARSAL=nccl1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
  bench.py --gpus 1 --steps 50 --warmup 10 --quick > $OUT/bench_nccl1.json 2> $OUT/bench_nccl1.err
echo "rc=$?"; tail -c 1800 $OUT/bench_nccl1.json; tail -5 $OUT/bench_nccl1.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
