"""Generates tests/golden/cemask_golden.npz: outputs of the reference's retrieve_masked_batch (trainers/CE.py:123-139)
run HERE on seeded inputs.  Only the function's text is exec'd in memory (importing the module would need TF);
nothing of the reference is written to disk -- the fixture holds inputs (brain masks, seed) and outputs (hole mask)."""
import random

import numpy as np

src = open('/root/reference/trainers/CE.py').read()
ns = {'np': np, 'random': random}
exec(src[src.index('def retrieve_masked_batch'):], ns)
ref = ns['retrieve_masked_batch']

cases = {}
for idx, (seed, n, h) in enumerate([(5, 6, 128), (11, 3, 64), (2, 1, 128), (7, 4, 32)]):
    rng = np.random.default_rng(seed)
    bm = np.zeros((n, h, h, 1), bool)
    for i in range(n):
        r0, c0 = rng.integers(2, h // 4, 2)
        r1, c1 = rng.integers(h // 2, h - 2, 2)
        bm[i, r0:r1, c0:c1] = True
    x = np.ones((n, h, h, 1))
    random.seed(seed)
    out = ref(x, bm)
    cases[f'bm{idx}'] = bm
    cases[f'seed{idx}'] = np.int64(seed)
    cases[f'holes{idx}'] = (out == 0)
np.savez_compressed('tests/golden/cemask_golden.npz', **cases)
print({k: v.shape for k, v in cases.items() if k.startswith('holes')}, [int(cases[f'holes{i}'].sum()) for i in range(4)])
