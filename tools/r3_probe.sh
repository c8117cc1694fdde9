#!/bin/bash
mkdir -p gpurun_out/r3
T="bn.gradfin final.gradfin dec3.wgrad dec2.wgrad dec0.wgrad enc3.wgrad enc1.wgrad dec0.dgrad bott.bwd enc3.dgrad enc1.dgrad enc0.fwd enc1.fwd adam"
for round in 1 2 3; do for cfg in "UAD_NO_REDUCE_NT=1" "UAD_X=1" "UAD_REDUCE4=1"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
UAD_REDUCE4=1 UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
tail -3 gpurun_out/r3/p.err
