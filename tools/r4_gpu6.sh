#!/bin/bash
# same-box sweep of UAD_SPLIT_TARGET (split-K of the small-spatial k5 layers): per-tag kernel times
export TMPDIR=/tmp
OUT=gpurun_out/r4_6; mkdir -p $OUT
T="enc3.fwd dec0.fwd dec1.fwd dec0.dgrad enc3.dgrad enc2.dgrad enc2.fwd dec1.dgrad"
for round in 1 2; do for S in 512 256 128 64; do
  UAD_SPLIT_TARGET=$S python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/b_$S.json 2>/dev/null
  echo -n "target=$S: "; python tools/kshow.py $OUT/b_$S.json $T
done; done
