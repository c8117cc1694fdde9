#!/bin/bash
# round-3 probe (second session): data-gradient kernels launched with the AQL barrier bit cleared (may start while the layer's filter gradient drains)
mkdir -p gpurun_out/r3
T="loss.finalize dec3.wgrad dec3.dgrad dec2.wgrad dec2.dgrad dec1.wgrad dec1.dgrad dec0.wgrad dec0.dgrad enc3.wgrad enc3.dgrad enc2.wgrad enc2.dgrad enc1.wgrad enc1.dgrad"
for round in 1 2 3; do for cfg in "UAD_X=1" "UAD_ANYORDER=1"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
UAD_ANYORDER=1 UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale_parity.py tests/test_gpu_trainers.py tests/test_gpu_cevae.py tests/test_gpu_dp_rehearsal.py -x -q 2>&1 | tail -2
tail -3 gpurun_out/r3/p.err
