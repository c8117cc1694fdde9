#!/bin/bash
# does an initialised RCCL process group slow the PLAIN step down?  same box, one process each: no process group | nccl | nccl with the watchdog features off | gloo | none again
export TMPDIR=/tmp
OUT=gpurun_out/r4_19; mkdir -p $OUT; L=$OUT/rccl_presence.log; : > $L
run() { echo "== $*" >> $L; env "$@" HT_ONLY_PLAIN=1 timeout 120 python tools/host_time_dp.py 2>&1 | grep "plain" >> $L; }
run HT_BACKEND=none
run HT_BACKEND=nccl
run HT_BACKEND=nccl HT_TOUCH=1
run HT_BACKEND=nccl HT_TOUCH=1 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_DUMP_ON_TIMEOUT=0
run HT_BACKEND=gloo
run HT_BACKEND=none
cat $L
