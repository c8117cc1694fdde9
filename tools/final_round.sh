#!/bin/bash
# On the GPU box: the round's closing evidence -- full GPU suite, default bench, rocprofv3 stats + PMC traffic passes -- into gpurun_out/final_<tag>/
TAG=${1:-r02}; COMMIT=${2:-unknown}
OUT=gpurun_out/final_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu_full.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
bash tools/profile_round.sh $TAG $COMMIT > $OUT/profile_round.log 2>&1
cp -r gpurun_out/prof_$TAG/* $OUT/ 2>/dev/null
tail -3 $OUT/pytest_gpu_full.log; head -c 600 $OUT/bench_default.json
