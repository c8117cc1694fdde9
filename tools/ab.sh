#!/bin/bash
# same-box A/B of two library builds (ablibs/libA.so vs ablibs/libB.so), interleaved rounds
T="dec3.fwd dec2.fwd dec1.fwd enc1.dgrad enc2.dgrad dec3.dgrad dec2.dgrad enc1.fwd enc2.fwd dec3.wgrad"
for round in 1 2 3; do for v in A B; do
  UAD_LIB=$PWD/ablibs/lib$v.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --math ${MATH:-bf16x3} > gpurun_out/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py gpurun_out/ab_$v.json $T
done; done
