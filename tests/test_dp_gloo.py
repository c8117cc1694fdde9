"""CPU, world_size 2 over gloo: the data-parallel contract.  Each rank computes the oracle's gradient on its shard of
the slice batch, the flat gradient buffer is all-reduced segment by segment exactly like
parallel.DataParallelStep does on RCCL, and sum/world must equal the gradient of the single-process big batch
(BatchNorm is frozen-affine, so DP is exact up to summation order).  Then one TF-Adam step keeps replicas identical."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nn as onn
from oracle import vae as ovae
from unsupervised_anomaly_detection_brain_mri_amd.parallel import allreduce_segments


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        h, inter, zdim, n = 32, 8, 16, 4
        m = ovae.Model('VAE', h, h, 1, inter, zdim)
        p = ovae.init_params(m.spec, seed=3, dtype=np.float64, perturb=True)
        x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float64)
        rng = np.random.default_rng(1)
        eps = rng.standard_normal((n, zdim))
        masks = {'mu': onn.make_dropout_mask(rng, (n, zdim), 0.2, np.float64),
                 'sigma': onn.make_dropout_mask(rng, (n, zdim), 0.2, np.float64),
                 'dec': onn.make_dropout_mask(rng, (n, 8 * 8 * 8), 0.2, np.float64)}
        per = n // world
        sl = slice(rank * per, (rank + 1) * per)
        lm = {k: v[sl] for k, v in masks.items()}
        out, cache = m.forward(p, x[sl], eps[sl], lm)
        g = m.backward(p, x[sl], out, cache, lm)
        flat = torch.from_numpy(ovae.flatten_params(m.spec, g).copy())
        # segments in completion order: decoder, bottleneck, encoder (flat layout is Encoder | Bottleneck | Decoder)
        sizes = {'Encoder': 0, 'Bottleneck': 0, 'Decoder': 0}
        for name, shape, _ in m.spec:
            sizes[name.split('/')[0]] += int(np.prod(shape))
        e, b, d = sizes['Encoder'], sizes['Bottleneck'], sizes['Decoder']
        works = allreduce_segments(flat, [(e + b, d), (e, b), (0, e)], world, async_op=True)
        for w in works:
            w.wait()
        flat /= world
        if rank == 0:
            out_f, cache_f = m.forward(p, x, eps, masks)
            g_f = ovae.flatten_params(m.spec, m.backward(p, x, out_f, cache_f, masks))
            err = np.abs(flat.numpy() - g_f).max() / np.abs(g_f).max()
            q.put(('grad_err', float(err)))
        # replicas stay identical after the optimizer step
        pf = ovae.flatten_params(m.spec, p).copy()
        mm, vv = np.zeros_like(pf), np.zeros_like(pf)
        onn.adam_tf_step(pf, flat.numpy(), mm, vv, 1, 1e-3, 0.5)
        t = torch.from_numpy(pf.copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        if rank == 0:
            q.put(('replica_diff', float((gathered[0] - gathered[1]).abs().max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_dp_equals_big_batch():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res['grad_err'] < 1e-12
    assert res['replica_diff'] == 0.0
