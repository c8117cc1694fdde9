#!/bin/bash
# timing ablations of the F5 kernel via UAD_DBG bits (1: no weight loads, 2: no halo staging, 4: no epilogue)
for d in 0 1 2 4 3 7; do
  UAD_DBG=$d UAD_BENCH_ALLOW_NAN=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | sed "s/NaN/0/g" > gpurun_out/ab_$d.json
  python - <<PY
import json
r=json.load(open("gpurun_out/ab_$d.json"))
k=r["kernels"]
print("dbg=$d", " ".join(f"{t}={k[t]['ms']*1e3:.0f}us/{k[t]['tflops']:.0f}TF" for t in ("enc1.fwd","enc2.fwd","enc3.fwd","dec3.dgrad","dec2.dgrad","dec1.dgrad")))
PY
done
