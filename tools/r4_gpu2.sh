#!/bin/bash
# round 4, k3 kernels (tap-list F / D + filter gradient): op-level parity in both modes, the ResNet model tests, and the configs[3] bench on one box
export TMPDIR=/tmp
OUT=gpurun_out/r4_2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops_resnet.py -q -m gpu --tb=short -s 2>&1 | grep -v "^$" | tail -80 > $OUT/ops_resnet.log; tail -4 $OUT/ops_resnet.log
timeout 900 python -m pytest tests/test_gpu_fanogan.py -q -m gpu -k "resnet" --tb=short -s 2>&1 | grep -v "^$" | grep -v "where \|+  " | tail -120 > $OUT/fanogan_resnet.log; tail -6 $OUT/fanogan_resnet.log
for M in bf16x3 bf16x3_all; do
  timeout 300 python bench.py --arch fAnoGAN --variant resnet --math $M --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_resnet_$M.json 2> $OUT/bench_resnet_$M.err
  python -c "import json,sys; r=json.load(open('$OUT/bench_resnet_$M.json')); print('$M', r['ms_per_step'], r['value'], r['config']['encoder_stage_ms_per_step'])" 2>&1 | tail -1
done
UAD_NO_X6=1 timeout 300 python bench.py --arch fAnoGAN --variant resnet --math bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_resnet_bf16x3_nox6.json 2> $OUT/bench_resnet_nox6.err
python -c "import json,sys; r=json.load(open('$OUT/bench_resnet_bf16x3_nox6.json')); print('bf16x3 UAD_NO_X6', r['ms_per_step'], r['value'], r['config']['encoder_stage_ms_per_step'])" 2>&1 | tail -1
