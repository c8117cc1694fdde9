#!/bin/bash
# round-3 probe (second session): two columns per thread in the slab reduction (8-byte streaming loads)
mkdir -p gpurun_out/r3
T="bn.gradfin dec3.wgrad dec2.dgrad dec0.dgrad enc3.wgrad enc1.dgrad"
for round in 1 2 3; do for cfg in "UAD_X=1" "UAD_REDUCE2=1"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
UAD_REDUCE2=1 UAD_MATH=bf16x3 timeout 600 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -2
