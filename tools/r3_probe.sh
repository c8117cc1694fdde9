#!/bin/bash
# round-3 probe (second session): occupancy throttle of the slab reduction (unused dynamic LDS per block)
mkdir -p gpurun_out/r3
T="bn.gradfin dec3.wgrad dec2.dgrad dec0.dgrad enc3.wgrad enc1.dgrad enc0.wgrad adam"
for round in 1 2 3; do for cfg in "UAD_X=1" "UAD_REDUCE_LDS=20000" "UAD_REDUCE_LDS=40000" "UAD_REDUCE_LDS=65000"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
tail -2 gpurun_out/r3/p.err
