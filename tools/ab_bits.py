"""Bit-identity check of two library builds: runs the same seeded VAE steps (configs[1] shape: 128 x 128, batch 64, dropout 0.2, explicit noise and
masks) on whichever library UAD_LIB names and prints sha256 digests of x_hat, of every gradient tensor group and of the parameters after three
optimizer steps.  `UAD_LIB=ablibs/libA.so python tools/ab_bits.py > a.txt; python tools/ab_bits.py > b.txt; diff a.txt b.txt` (tools/r6_gpu.sh bits)."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn as onn, vae as ovae          # synthetic inputs only (tools/ may use the oracle's data helpers)
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine


def main():
    math = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
    arch = sys.argv[2] if len(sys.argv) > 2 else 'VAE'
    n, h, zdim = 64, 128, 128
    m = ovae.Model(arch, h, h, 1, 8, zdim)
    p = ovae.init_params(m.spec, seed=3, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=0)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((n, zdim)).astype(np.float32)
    masks = {'mu': onn.make_dropout_mask(rng, (n, zdim), 0.2), 'sigma': onn.make_dropout_mask(rng, (n, zdim), 0.2),
             'dec': onn.make_dropout_mask(rng, (n, 8 * 8 * 16), 0.2)}
    eng = Engine(arch, h, h, 1, 8, zdim, max_batch=n, device='cuda:0')
    eng.set_math(math)
    eng.set_params(p)
    dig = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    for step in range(3):
        out = eng.train_step(x, eps, masks, lr=1e-4)
        torch.cuda.synchronize()
        g = eng.get_grads()
        print(f'step {step} x_hat {dig(out["x_hat"].cpu().numpy())} scalars {dig(out["scalars"].cpu().numpy())}')
        for k in sorted(g):
            print(f'  grad {k:24s} {dig(g[k])}')
    print('params', dig(eng.get_buffer_host()))
    eng.close()


if __name__ == '__main__':
    main()
