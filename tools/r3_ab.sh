#!/bin/bash
# round-3 A/B on one box: the op / model parity tests with the lane = pixel kernels on, then interleaved bench rounds with env switches
# usage: tools/r3_ab.sh "<envA>" "<envB>" [tests...]
A="$1"; B="$2"; shift 2
mkdir -p gpurun_out/r3
if [ $# -gt 0 ]; then
  UAD_MATH=bf16x3 timeout 900 python -m pytest "$@" -x -q 2>&1 | tail -15 > gpurun_out/r3/pytest.log; cat gpurun_out/r3/pytest.log
fi
T="enc1.fwd enc2.fwd enc3.fwd dec0.fwd dec1.fwd dec2.fwd dec3.fwd dec3.wgrad dec3.dgrad dec2.wgrad dec2.dgrad dec1.dgrad dec0.dgrad enc3.dgrad enc2.dgrad enc1.dgrad enc1.wgrad enc2.wgrad"
for round in ${ROUNDS:-1 2 3}; do
  env $A python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/ab_A.json 2>gpurun_out/r3/ab_A.err
  echo -n "A[$A]: "; python tools/kshow.py gpurun_out/r3/ab_A.json $T
  env $B python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/ab_B.json 2>gpurun_out/r3/ab_B.err
  echo -n "B[$B]: "; python tools/kshow.py gpurun_out/r3/ab_B.json $T
done
tail -3 gpurun_out/r3/ab_B.err
