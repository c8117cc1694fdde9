#!/bin/bash
mkdir -p gpurun_out/r3
UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_scale_parity.py -x -q 2>&1 | tail -3
T="dec3.wgrad dec2.wgrad dec1.wgrad dec0.wgrad enc3.wgrad enc2.wgrad enc1.wgrad"
for round in 1 2 3; do for cfg in "UAD_NO_W_TW8=1" "UAD_X=1"; do
  env $cfg python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/p.json 2>/dev/null
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
