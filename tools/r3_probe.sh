#!/bin/bash
# round-3 probe (second session): bf16 weight repack with one 16-byte group per thread (UAD_NO_PACK8 = the per-element kernel)
mkdir -p gpurun_out/r3
UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_optimizers.py tests/test_gpu_fanogan.py -x -q 2>&1 | tail -2
T="pack.weights enc0.fwd enc1.fwd enc2.fwd adam"
for round in 1 2 3; do for cfg in "UAD_NO_PACK8=1" "UAD_X=1"; do
  env $cfg python bench.py --steps 60 --warmup 10 --quick --rounds 3 > gpurun_out/r3/p.json 2>gpurun_out/r3/p.err
  echo -n "[$cfg]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done; done
tail -2 gpurun_out/r3/p.err
