#!/bin/bash
mkdir -p gpurun_out/r3
UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_cevae.py -x -q 2>&1 | tail -3
T="enc1.fwd enc2.fwd dec2.fwd dec3.fwd dec3.dgrad dec2.dgrad enc1.dgrad dec3.wgrad dec2.wgrad dec1.wgrad dec0.wgrad enc3.wgrad enc2.wgrad enc1.wgrad"
for round in 1 2 3; do for v in A B; do
  UAD_LIB=$PWD/ablibs/lib$v.so python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py gpurun_out/r3/ab_$v.json $T
done; done
