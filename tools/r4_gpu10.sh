#!/bin/bash
# transpose-read filter gradient (UAD_W_TR=1): op-level + model parity, then same-box A/B of the wgrad tags
export TMPDIR=/tmp
OUT=gpurun_out/r4_10; mkdir -p $OUT
UAD_W_TR=1 UAD_MATH=bf16x3 timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv_w" --tb=short 2>&1 | grep -v "^$" | tail -15 > $OUT/ops.log; tail -3 $OUT/ops.log
UAD_W_TR=1 timeout 900 python -m pytest tests/test_gpu_ops_large.py tests/test_gpu_model.py tests/test_gpu_scale_parity.py -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -15 > $OUT/model.log; tail -3 $OUT/model.log
T="dec3.wgrad dec2.wgrad dec1.wgrad dec0.wgrad enc3.wgrad enc2.wgrad enc1.wgrad"
for round in 1 2 3; do for v in base tr; do
  if [ $v = tr ]; then export UAD_W_TR=1; else unset UAD_W_TR; fi
  python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py $OUT/ab_$v.json $T
done; done
