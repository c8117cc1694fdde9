#!/bin/bash
# round-3 probe (second session): any-order data-gradient launches in the exact-fp32 mode too; the request no longer outlives a launch that took no spatial kernel
mkdir -p gpurun_out/r3
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_shapes.py -x -q 2>&1 | tail -2
for round in 1 2; do for cfg in "UAD_NO_ANYORDER=1" "UAD_X=1"; do
  env $cfg python bench.py --steps 40 --warmup 5 --quick --rounds 3 --math f32 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[f32 $cfg]: "; python tools/kshow.py gpurun_out/r3/p.json dec3.dgrad enc1.dgrad
done; done
