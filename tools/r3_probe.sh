#!/bin/bash
# round-3 probe (second session): loss.finalize of a training step on the side stream
mkdir -p gpurun_out/r3
UAD_MATH=bf16x3 timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale_parity.py tests/test_gpu_trainers.py tests/test_gpu_cevae.py tests/test_gpu_dp_rehearsal.py tests/test_gpu_spatial_ae.py tests/test_gpu_gmvae.py -x -q 2>&1 | tail -3
T="loss.finalize bn.gradfin final.gradfin dec3.fwd dec3.wgrad dec3.dgrad adam"
for round in 1 2 3; do for cfg in "UAD_LOSS_ON_MAIN=1" "UAD_X=1"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
python bench.py --arch ceVAE --steps 30 --warmup 5 --quick 2>/dev/null | cut -c1-300
tail -3 gpurun_out/r3/p.err
