#!/bin/bash
# E3 (uad_backward_adam: early optimizer update + repack on the side stream): tests, then same-box A/B against UAD_NO_EARLY_ADAM=1
export TMPDIR=/tmp
OUT=gpurun_out/r4_9; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_optimizers.py tests/test_gpu_trainers.py tests/test_gpu_model.py tests/test_gpu_dp_rehearsal.py tests/test_gpu_dp_nccl.py -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -30 > $OUT/tests.log; tail -5 $OUT/tests.log
for round in 1 2 3; do for v in early plain; do
  if [ $v = plain ]; then export UAD_NO_EARLY_ADAM=1; else unset UAD_NO_EARLY_ADAM; fi
  python bench.py --steps 100 --warmup 10 --quick --rounds 3 > $OUT/ab_$v.json 2>/dev/null
  python -c "import json; r=json.load(open('$OUT/ab_$v.json')); print('$v', r['ms_per_step'], r['value'], r.get('trainer_loop_slices_per_s'))"
done; done
