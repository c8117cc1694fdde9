#!/bin/bash
# what about an RCCL communicator slows the plain step (no collective issued) by ~17 %?  same box, one process each
export TMPDIR=/tmp
OUT=gpurun_out/r4_20; mkdir -p $OUT; L=$OUT/rccl_presence2.log; : > $L
run() { echo "== $*" >> $L; env "$@" HT_ONLY_PLAIN=1 timeout 120 python tools/host_time_dp.py 2>&1 | grep "plain" >> $L; }
run HT_BACKEND=none
run HT_BACKEND=nccl HT_LAZY=1
run HT_BACKEND=nccl HT_LAZY=1 HT_TOUCH=1
run HT_BACKEND=nccl HT_ENGINE_FIRST=1
run HT_BACKEND=nccl UAD_NO_ANYORDER=1
run HT_BACKEND=nccl UAD_EVENT_SYSFENCE=1
run HT_BACKEND=nccl NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
run HT_BACKEND=nccl GPU_MAX_HW_QUEUES=8
run HT_BACKEND=nccl HSA_ENABLE_INTERRUPT=0
run HT_BACKEND=none UAD_NO_ANYORDER=1
cat $L
