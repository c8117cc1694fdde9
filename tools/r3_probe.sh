#!/bin/bash
# round-3 probe (second session): the other workloads on the round's final build (streaming slab reads + any-order launches are shared by all handles)
mkdir -p gpurun_out/r3
python bench.py --arch ceVAE --steps 40 --warmup 5 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ceVAE16', d['value'], d['ms_per_step'])"
python bench.py --arch ceVAE --batch 64 --steps 40 --warmup 5 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ceVAE64', d['value'], d['ms_per_step'])"
python bench.py --arch GMVAE_spatial --steps 3 --warmup 1 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GMVAE_spatial', d['value'], d['ms_per_step'], d['config'].get('workload','')[:80])"
python bench.py --arch fAnoGAN --steps 10 --warmup 2 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fAnoGAN unified', d['value'], d['ms_per_step'])"
python bench.py --arch fAnoGAN --variant resnet --steps 5 --warmup 2 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fAnoGAN resnet', d['value'], d['ms_per_step'])"
