#!/bin/bash
# transpose-read filter gradient: sweep of the workgroup-slab target (UAD_W5_TARGET), same box -- the STEP time is what counts: the layer's data gradient is
# launched behind the filter gradient without the barrier bit and fills the CU slots the filter gradient leaves free
export TMPDIR=/tmp
OUT=gpurun_out/r4_14; mkdir -p $OUT
T="dec3.wgrad dec3.dgrad enc2.wgrad enc2.dgrad"
for round in 1 2; do for S in 512 448 384 320 256 192; do
  UAD_W5_TARGET=$S python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/b_$S.json 2>/dev/null
  echo -n "target=$S: "; python tools/kshow.py $OUT/b_$S.json $T
done; done
