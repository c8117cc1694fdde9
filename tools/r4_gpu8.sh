#!/bin/bash
# E1 (gather kernel with the lane = pixel epilogue): parity subset on the new build, then same-box A/B (ablibs/libA.so = before, libB.so = after)
export TMPDIR=/tmp
OUT=gpurun_out/r4_8; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_large.py tests/test_gpu_model.py tests/test_gpu_scale_parity.py tests/test_gpu_cevae.py tests/test_gpu_gmvae.py -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -30 > $OUT/tests.log; tail -4 $OUT/tests.log
T="enc1.fwd enc2.fwd enc3.fwd dec3.dgrad dec2.dgrad dec1.dgrad dec0.dgrad"
for round in 1 2 3; do for v in A B; do
  UAD_LIB=$PWD/ablibs/lib$v.so python bench.py --steps 50 --warmup 5 --quick --rounds 3 > $OUT/ab_$v.json 2>/dev/null
  echo -n "$v: "; python tools/kshow.py $OUT/ab_$v.json $T
done; done
