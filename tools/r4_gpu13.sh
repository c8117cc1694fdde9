#!/bin/bash
# transpose-read filter gradient: 64-column cs blocks on eight waves (default) vs 32-column blocks on four waves, two workgroups per CU (UAD_NO_W2=1)
export TMPDIR=/tmp
OUT=gpurun_out/r4_13; mkdir -p $OUT
T="dec3.wgrad dec2.wgrad dec1.wgrad dec0.wgrad enc3.wgrad enc2.wgrad enc1.wgrad"
for round in 1 2; do for v in pair single; do
  if [ $v = single ]; then export UAD_NO_W2=1; else unset UAD_NO_W2; fi
  python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py $OUT/ab_$v.json $T
done; done
