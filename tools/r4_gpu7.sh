#!/bin/bash
# after the per-kind split target: model-level parity subset + bench in both modes (same box as each other)
export TMPDIR=/tmp
OUT=gpurun_out/r4_7; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale_parity.py tests/test_gpu_ops.py tests/test_gpu_ops_large.py tests/test_gpu_cevae.py tests/test_gpu_gmvae.py tests/test_gpu_shapes.py -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -30 > $OUT/tests.log; tail -4 $OUT/tests.log
for S in 512 0; do
  if [ $S = 0 ]; then unset UAD_SPLIT_TARGET; else export UAD_SPLIT_TARGET=$S; fi
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --rounds 3 > $OUT/b_$S.json 2>/dev/null
  python - $OUT/b_$S.json $S <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); o = r.get('other_math_mode', {})
print('split', sys.argv[2] or 'default', 'bf16x3', r['ms_per_step'], r['value'], '| f32', o.get('ms_per_step'), o.get('value'))
PY
done
