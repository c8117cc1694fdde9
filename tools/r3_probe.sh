#!/bin/bash
mkdir -p gpurun_out/r3
for cfg in "UAD_DBG=$((64+256*40)) UAD_STAGGER=512"; do
  echo "== $cfg"; env $cfg python bench.py --steps 60 --warmup 3 --quick --rounds 1 2>&1 >/dev/null | grep d16s | head -2
done
T="dec0.fwd dec1.fwd dec2.fwd dec3.fwd enc3.dgrad enc2.dgrad enc1.dgrad"
for round in 1 2; do
for cfg in "UAD_X=1" "UAD_STAGGER=512" "UAD_NO_PP=1"; do
  env $cfg python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/p.json 2>/dev/null
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
