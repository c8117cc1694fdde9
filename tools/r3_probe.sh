#!/bin/bash
mkdir -p gpurun_out/r3
UAD_MATH=bf16x3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_scale_parity.py tests/test_gpu_cevae.py -x -q 2>&1 | tail -5
T="enc1.fwd enc2.fwd enc3.fwd dec3.dgrad dec2.dgrad dec1.dgrad dec0.dgrad"
for round in 1 2; do for v in A B; do
  UAD_LIB=$PWD/ablibs/lib$v.so python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py gpurun_out/r3/ab_$v.json $T
done; done
for tp in 1 2 4 8; do
  UAD_F16_TPW=$tp python bench.py --steps 40 --warmup 5 --quick > gpurun_out/r3/p.json 2>/dev/null
  echo -n "[tpw $tp]: "; python tools/kshow.py gpurun_out/r3/p.json $T
done
