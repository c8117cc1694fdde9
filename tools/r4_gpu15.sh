#!/bin/bash
# rehearsal of the driver's multi-GPU launch line on the one GPU (two ranks on cuda:0, gloo): bench.py --gpus 2 must still produce its JSON line
export TMPDIR=/tmp
OUT=gpurun_out/r4_15; mkdir -p $OUT
UAD_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2.err
tail -c 1500 $OUT/bench_gpus2_rehearsal.json; echo; tail -3 $OUT/bench_gpus2.err
