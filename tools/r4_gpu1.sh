#!/bin/bash
# round 4, first GPU call: the new tests, then the full GPU suite, then the default bench line (no CPU baseline)
export TMPDIR=/tmp
OUT=gpurun_out/r4_1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dp_nccl.py tests/test_gpu_knobs.py -x -q -m gpu 2>&1 | tail -25 > $OUT/new_tests.log
tail -3 $OUT/new_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_knobs.py --deselect tests/test_gpu_dp_nccl.py 2>&1 | tail -25 > $OUT/pytest_gpu_full.log
tail -3 $OUT/pytest_gpu_full.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo
