#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4_3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fanogan.py -q -m gpu -k "resnet" --tb=short -s 2>&1 | grep -v "^$" | tail -150 > $OUT/fanogan_resnet.log; tail -8 $OUT/fanogan_resnet.log
