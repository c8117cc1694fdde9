"""Table of an A/B directory written by tools/r5_gpu.sh (files <math>_<A|B>_<round>.json = bench.py lines): ms per step and the per-tag
kernel times (us) of every round, A = ablibs/libA.so, B = the tree's library."""
import glob
import json
import os
import sys

d = sys.argv[1]
tags = sys.argv[2:] or ['dec3.fwd', 'dec3.dgrad', 'dec3.wgrad', 'dec2.wgrad', 'dec1.wgrad', 'dec0.wgrad', 'enc1.wgrad', 'enc2.wgrad', 'enc3.wgrad', 'enc1.fwd', 'dec2.fwd']
for f in sorted(glob.glob(os.path.join(d, '*_[AB]_*.json'))):
    try:
        j = json.load(open(f))
        k = j['kernels']
        w = sum(k[t]['ms'] for t in k if t.endswith('.wgrad') and t not in ('enc0.wgrad', 'bott.wgrad')) * 1e3
        print(f"{os.path.basename(f):22s} {j['ms_per_step']:.4f} ms  {j['value']:9.1f}/s  sum k5 wgrad {w:6.1f} | " + ' '.join(f"{t}={k[t]['ms'] * 1e3:.1f}" for t in tags if t in k))
    except Exception as e:
        print(f, 'failed:', e)
