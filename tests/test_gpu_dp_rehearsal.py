"""GPU, two ranks sharing the one device, gloo standing in for RCCL: the data-parallel TRAIN step of the real engines -- gradient
all-reduce over the handle-owned buffers, Adam with grad_scale = 1 / world -- equals the single-process step on the concatenated batch
(frozen-stats BN / per-sample LayerNorm make the per-sample gradients independent; the only difference is fp32 summation order).
Covers the fused VAE handle (segmented async all-reduce), the dense GMVAE trainer (inline all-reduce of the 'AE' group), the Zimmerer VAE
(Engine adapter under DataParallelStep) and the AAE family on residual blocks (per-phase group all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cases():
    from unsupervised_anomaly_detection_brain_mri_amd import models, trainers
    return [('VAE', trainers.VAE, models.variational_autoencoder, dict(zDim=16), [8, 8], 32),
            ('GMVAE', trainers.GMVAE, models.gaussian_mixture_variational_autoencoder, dict(zDim=16), [8, 8], 32),
            ('VAE_Zimmerer', trainers.VAE, models.variational_autoencoder_Zimmerer, dict(zDim=16), [2, 2], 32),
            ('CAAE_Chen', trainers.ConstrainedAAE, models.constrained_adversarial_autoencoder_Chen, dict(zDim=16), [4, 4], 32)]


def _make(case, world, tmp, bs):
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    name, T, net, kw, inter, h = case
    opt = get_options(batchsize=bs, learningrate=1e-3, numEpochs=1, outputWidth=h, outputHeight=h,
                      config={'CHECKPOINTDIR': os.path.join(tmp, 'ck'), 'SAMPLEDIR': os.path.join(tmp, 'smp')}, **kw)
    ds = SyntheticDataset(8, 8, h, h, seed=0)
    cfg = get_config(T, opt, 'ADAM', inter, 0.0, ds)          # dropout rate 0: no masks to keep in step between the two runs
    return T(None, cfg, network=net, seed=7, world=world, device='cuda:0')


def _train_step(name, model, x, rng):
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import Phase
    n = len(x)
    if name == 'GMVAE':
        eps = (rng.standard_normal((n, 1)).astype(np.float32), rng.standard_normal((n, 1)).astype(np.float32))
        return eps, lambda m, xb, e: m.step(xb, Phase.TRAIN, eps=e, fetch_maps=False)
    if name == 'CAAE_Chen':
        return None, lambda m, xb, e: m.step(xb, Phase.TRAIN, fetch_maps=False)
    eps = rng.standard_normal((n, 16)).astype(np.float32)
    return eps, lambda m, xb, e: m.step(xb, Phase.TRAIN, eps=e, fetch_maps=False)


def _worker(rank, world, port, tmp, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unsupervised_anomaly_detection_brain_mri_amd import _lib
        from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
        for case in _cases():
            name = case[0]
            n = 4
            x = synthetic_slices(n, case[5], case[5], seed=3)
            rng = np.random.default_rng(11)
            dpm = _make(case, world, os.path.join(tmp, f'dp{rank}'), n // world)
            eps, run = _train_step(name, dpm, x, rng)
            sl = slice(rank * (n // world), (rank + 1) * (n // world))
            e_loc = None if eps is None else (tuple(e[sl] for e in eps) if isinstance(eps, tuple) else eps[sl])
            w0 = dpm.engine.get_buffer_host(_lib.BUF_PARAMS)
            r_dp = run(dpm, x[sl], e_loc)
            w_dp = dpm.engine.get_buffer_host(_lib.BUF_PARAMS)
            if rank == 0:
                single = _make(case, 1, os.path.join(tmp, 'single'), n)
                single.engine.set_params(w0)                       # same start as the (broadcast) replicas
                single.engine.reset_optimizer()
                r_1 = run(single, x, eps)
                w_1 = single.engine.get_buffer_host(_lib.BUF_PARAMS)
                moved = np.abs(w_1 - w0)
                q.put((name, dict(loss_dp=float(r_dp['loss']), loss_1=float(r_1['loss']),
                                  mean_diff=float(np.abs(w_dp - w_1).mean()), mean_step=float(moved.mean()),
                                  frac_far=float((np.abs(w_dp - w_1) > 0.5 * 1e-3).mean()))))
                single.engine.close()
            t = torch.from_numpy(w_dp)
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t)
            if rank == 0:
                q.put((name + '_replicas', float((both[0] - both[1]).abs().max())))
            dpm.engine.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_equals_big_batch(tmp_path):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = {}
    while not q.empty():
        k, v = q.get()
        res[k] = v
    for name in ('VAE', 'GMVAE', 'VAE_Zimmerer', 'CAAE_Chen'):
        r = res[name]
        assert res[name + '_replicas'] == 0.0, name                       # both replicas applied the identical update
        # the reported loss is the all-reduced mean of the two half-batch means = the big-batch mean
        assert r['loss_dp'] == pytest.approx(r['loss_1'], rel=2e-4), (name, r)
        # one Adam step moves a weight by ~lr * sign(g): the DP and big-batch runs agree except where a gradient is rounding noise
        assert r['mean_diff'] <= 0.02 * r['mean_step'] and r['frac_far'] <= 0.01, (name, r)
