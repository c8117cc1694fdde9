#!/bin/bash
# transpose-read filter gradient: double-buffered tile loop (UAD_W_DB=1) vs single buffer, same box; parity of the DB form first
export TMPDIR=/tmp
OUT=gpurun_out/r4_11; mkdir -p $OUT
UAD_W_TR=1 UAD_W_DB=1 UAD_MATH=bf16x3 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_large.py -q -m gpu -k "conv_w or wgrad or filter" --tb=short 2>&1 | grep -v "^$" | tail -8 > $OUT/ops.log; tail -2 $OUT/ops.log
UAD_W_TR=1 UAD_W_DB=1 timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -8 > $OUT/model.log; tail -2 $OUT/model.log
T="dec3.wgrad dec2.wgrad dec1.wgrad dec0.wgrad enc3.wgrad enc2.wgrad enc1.wgrad"
export UAD_W_TR=1
for round in 1 2 3; do for v in tr trdb; do
  if [ $v = trdb ]; then export UAD_W_DB=1; else unset UAD_W_DB; fi
  python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py $OUT/ab_$v.json $T
done; done
