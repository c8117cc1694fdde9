"""CPU, world_size 2 over gloo: the data-parallel contract.  Each rank computes the oracle's gradient on its shard of
the slice batch, the flat gradient buffer is all-reduced segment by segment exactly like
parallel.DataParallelStep does on RCCL, and sum/world must equal the gradient of the single-process big batch
(BatchNorm is frozen-affine, so DP is exact up to summation order).  Then one TF-Adam step keeps replicas identical."""
import os
import socket

import numpy as np
import pytest
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


class _OracleEngine:
    """The Engine surface parallel.DataParallelStep drives (forward / backward(segment) / grad_segment / buffer / adam_step), computed by the
    oracle on CPU: backward(segment) fills ONLY that segment's slice of the flat gradient buffer (in the handle's order DECODER, BOTTLENECK,
    ENCODER_HI, ENCODER_LO; include/uad_hip.h), so an all-reduce issued too early, on the wrong slice, or a slice left out shows up as a wrong
    gradient.  The flat layout is the handle's: Encoder | Bottleneck | Decoder, ENCODER_LO = [enc0.kernel .. enc1.kernel]."""

    def __init__(self, model, params64):
        from unsupervised_anomaly_detection_brain_mri_amd import _lib
        self._lib, self.m, self.p = _lib, model, params64
        self.spec = model.spec
        self.nparams = sum(int(np.prod(sh)) for _, sh, _ in self.spec)
        self.grads = torch.zeros(self.nparams, dtype=torch.float64)
        self.flat = ovae.flatten_params(self.spec, params64).copy()
        self.mm, self.vv, self.t = np.zeros_like(self.flat), np.zeros_like(self.flat), 0
        off, acc = {}, 0
        for name, sh, _ in self.spec:            # flatten_params order = spec order
            off[name] = acc
            acc += int(np.prod(sh))
        size = {'Encoder': 0, 'Bottleneck': 0, 'Decoder': 0}
        for name, sh, _ in self.spec:
            size[name.split('/')[0]] += int(np.prod(sh))
        e, b, d = size['Encoder'], size['Bottleneck'], size['Decoder']
        n_enc = sum(1 for name, _, _ in self.spec if name.startswith('Encoder/enc_conv2D_') and name.endswith('kernel'))
        split = off['Encoder/enc_conv2D_1/bias'] if n_enc >= 3 else e
        self.segs = {_lib.SEG_DECODER: (e + b, d), _lib.SEG_BOTTLENECK: (e, b), _lib.SEG_ENCODER: (0, e),
                     _lib.SEG_ENCODER_LO: (0, split), _lib.SEG_ENCODER_HI: (split, e - split)}
        self.calls = []

    def grad_segment(self, seg):
        return self.segs[seg]

    def buffer(self, which):
        assert which == self._lib.BUF_GRADS
        return self.grads

    def forward(self, x, eps=None, masks=None, want_backward=False, **kw):
        out, cache = self.m.forward(self.p, x, eps, masks)
        self._full = torch.from_numpy(ovae.flatten_params(self.spec, self.m.backward(self.p, x, out, cache, masks)).copy())
        self.grads.fill_(float('nan'))            # nothing is valid until its segment's backward ran
        return {'scalars': torch.zeros(8)}

    def backward(self, seg):
        self.calls.append(seg)
        off, cnt = self.segs[seg]
        self.grads[off:off + cnt] = self._full[off:off + cnt]

    def backward_deferred(self, seg):
        """Engine.backward_deferred's contract on a host without streams: the slice is complete in the caller's order -> None (the device engine
        names its side stream for the first three segments; tests/test_gpu_model.py and tests/test_gpu_dp_nccl.py cover that form)."""
        self.deferred_calls = getattr(self, 'deferred_calls', 0) + 1
        self.backward(seg)
        return None

    def adam_step(self, lr, beta1, beta2, eps, grad_scale):
        self.t += 1
        onn.adam_tf_step(self.flat, self.grads.numpy() * grad_scale, self.mm, self.vv, self.t, lr, beta1, beta2, eps)


def _dp_worker(rank, world, port, q, h, buckets=4):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unsupervised_anomaly_detection_brain_mri_amd import _lib
        from unsupervised_anomaly_detection_brain_mri_amd.parallel import SEGMENT_ORDER, DataParallelStep
        inter, zdim, n = 8, 16, 4
        m = ovae.Model('VAE', h, h, 1, inter, zdim)
        p = ovae.init_params(m.spec, seed=3, dtype=np.float64, perturb=True)
        x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float64)
        rng = np.random.default_rng(1)
        eps = rng.standard_normal((n, zdim))
        flat_dim = 8 * 8 * p['Bottleneck/conv2d/kernel'].shape[-1]
        masks = {'mu': onn.make_dropout_mask(rng, (n, zdim), 0.2, np.float64), 'sigma': onn.make_dropout_mask(rng, (n, zdim), 0.2, np.float64),
                 'dec': onn.make_dropout_mask(rng, (n, flat_dim), 0.2, np.float64)}
        per = n // world
        sl = slice(rank * per, (rank + 1) * per)
        eng = _OracleEngine(m, p)
        dp = DataParallelStep(eng, world, buckets=buckets)
        assert len(dp.plan) == min(buckets, 4) and sum(c for _, _, c in dp.plan) == eng.nparams
        dp.train_step(x[sl], eps[sl], {k: v[sl] for k, v in masks.items()}, lr=1e-3, beta1=0.5)
        assert tuple(eng.calls) == SEGMENT_ORDER == (_lib.SEG_DECODER, _lib.SEG_BOTTLENECK, _lib.SEG_ENCODER_HI, _lib.SEG_ENCODER_LO)
        assert sorted(dp.segs) == sorted(SEGMENT_ORDER) and sum(c for _, c in dp.segs.values()) == eng.nparams
        if rank == 0:
            ref = _OracleEngine(m, p)
            ref.forward(x, eps, masks)
            g_f = ref._full.numpy()
            got = eng.grads.numpy() / world
            q.put(('grad_err', float(np.abs(got - g_f).max() / np.abs(g_f).max())))
            q.put(('hi_share', float(dp.segs[_lib.SEG_ENCODER_HI][1]) / max(1, eng.segs[_lib.SEG_ENCODER][1])))
        t = torch.from_numpy(eng.flat.copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        if rank == 0:
            q.put(('replica_diff', float((gathered[0] - gathered[1]).abs().max())))
    finally:
        dist.destroy_process_group()


def _run_dp(h, buckets=4):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, h, buckets)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    return dict(q.get(timeout=5) for _ in range(3))


def test_data_parallel_step_issues_the_four_segments():
    """parallel.DataParallelStep.train_step itself, two ranks over gloo, on an oracle-backed engine: DECODER, BOTTLENECK, ENCODER_HI,
    ENCODER_LO in that order, each all-reduced after its backward; the all-reduced sum / world is the big-batch gradient and the replicas stay
    identical after Adam.  64 x 64 has three encoder blocks (ENCODER_HI non-empty), 32 x 32 two (ENCODER_HI empty: skipped)."""
    res = _run_dp(64)
    assert res['grad_err'] < 1e-12 and res['replica_diff'] == 0.0 and 0.5 < res['hi_share'] < 1.0
    res = _run_dp(32)
    assert res['grad_err'] < 1e-12 and res['replica_diff'] == 0.0 and res['hi_share'] == 0.0


@pytest.mark.parametrize('buckets', [3, 2, 1])
def test_data_parallel_step_bucket_plans(buckets):
    """UAD_DP_BUCKETS: the four gradient segments merged into 3 / 2 / 1 all-reduce calls (parallel.bucket_plan).  A merged slice is reduced
    only after the LAST of its segments is back-propagated (the oracle-backed engine leaves NaN in every slice whose backward has not run, so
    an early collective poisons the gradient), every element is reduced exactly once, and the result is the big-batch gradient."""
    res = _run_dp(64, buckets)
    assert res['grad_err'] < 1e-12 and res['replica_diff'] == 0.0


def test_bucket_plan_and_no_allreduce_switch():
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep, bucket_plan
    segs = {_lib.SEG_ENCODER_LO: (0, 50), _lib.SEG_ENCODER_HI: (50, 600), _lib.SEG_BOTTLENECK: (650, 400), _lib.SEG_DECODER: (1050, 700)}
    assert bucket_plan(segs, 4) == [(_lib.SEG_DECODER, 1050, 700), (_lib.SEG_BOTTLENECK, 650, 400), (_lib.SEG_ENCODER_HI, 50, 600), (_lib.SEG_ENCODER_LO, 0, 50)]
    assert bucket_plan(segs, 3) == [(_lib.SEG_DECODER, 1050, 700), (_lib.SEG_ENCODER_HI, 50, 1000), (_lib.SEG_ENCODER_LO, 0, 50)]
    assert bucket_plan(segs, 2) == [(_lib.SEG_ENCODER_HI, 50, 1700), (_lib.SEG_ENCODER_LO, 0, 50)]
    assert bucket_plan(segs, 1) == [(_lib.SEG_ENCODER_LO, 0, 1750)]
    # the spatial AE has no bottleneck variables: empty segments drop out of a merged slice
    segs[_lib.SEG_BOTTLENECK] = (650, 0); segs[_lib.SEG_DECODER] = (650, 700)
    assert bucket_plan(segs, 3)[1] == (_lib.SEG_ENCODER_HI, 50, 600)
    with pytest.raises(ValueError):
        bucket_plan(segs, 5)


# ---------------------------------------------------------------------------------------------------------------
# f-AnoGAN phases under DP: parallel.GanDataParallel itself (all-reduce of the trained group's slice, Adam with
# grad_scale 1/world) driven over gloo with an oracle-backed stand-in for the device engine.
# ---------------------------------------------------------------------------------------------------------------
class _OracleGanEngine:
    def __init__(self, m, p):
        from oracle import fanogan as ofa
        self.m, self.ofa = m, ofa
        self.spec, off = [], 0
        for name, shape, _ in m.spec:
            self.spec.append((name, shape, off)); off += int(np.prod(shape))
        self.params = torch.from_numpy(np.concatenate([p[k].reshape(-1) for k, _, _ in self.spec]).copy())
        self.grads = torch.zeros_like(self.params)
        self.m1, self.m2 = np.zeros(off), np.zeros(off)
        self.t = {'Encoder': 0, 'Generator': 0, 'Discriminator': 0}

    def buffer(self, which):
        return self.params if which == 0 else self.grads

    def group(self, g):
        idx = [(o, int(np.prod(s))) for k, s, o in self.spec if self.ofa.group_of(k) == g]
        return idx[0][0], sum(c for _, c in idx)

    def _p(self):
        flat = self.params.numpy()
        return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, s, o in self.spec}

    def phase(self, group, want_backward=True, x=None, z=None, alpha=None, **kw):
        p = self._p()
        if group == 'Generator':
            ls, g = self.m.gen_phase(p, z)
        elif group == 'Discriminator':
            ls, g = self.m.disc_phase(p, x, z, alpha)
        else:
            ls, g = self.m.enc_phase(p, x)
        for k, s, o in self.spec:
            if k in g:
                self.grads[o:o + int(np.prod(s))] = torch.from_numpy(np.asarray(g[k], np.float64).reshape(-1).copy())
        return ls

    def adam(self, group, lr, b1, b2, eps, grad_scale):
        self.t[group] += 1
        off, cnt = self.group(group)
        sl = slice(off, off + cnt)
        pf = self.params.numpy()
        onn.adam_tf_step(pf[sl], self.grads.numpy()[sl] * grad_scale, self.m1[sl], self.m2[sl], self.t[group], lr, b1, b2, eps)


def _gan_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import fanogan as ofa
        from unsupervised_anomaly_detection_brain_mri_amd.parallel import GanDataParallel
        h, inter, zdim, n = 32, 8, 16, 4
        m = ofa.FAnoGAN(h, inter, zdim)
        p = ovae.init_params(m.spec, seed=4, dtype=np.float64, perturb=True)
        x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float64)
        rng = np.random.default_rng(2)
        z = rng.standard_normal((n, zdim)); alpha = rng.uniform(0, 1, (n, 1))
        per = n // world
        sl = slice(rank * per, (rank + 1) * per)
        eng = _OracleGanEngine(m, p)
        dp = GanDataParallel(eng, world)
        single = _OracleGanEngine(m, p)
        for group, kw, kw_full in (('Discriminator', dict(x=x[sl], z=z[sl], alpha=alpha[sl]), dict(x=x, z=z, alpha=alpha)),
                                   ('Generator', dict(z=z[sl]), dict(z=z)), ('Encoder', dict(x=x[sl]), dict(x=x))):
            dp.train_phase(group, 1e-3, **kw)
            single.phase(group, **kw_full)
            off, cnt = eng.group(group)
            g_dp = eng.grads[off:off + cnt].numpy() / world
            g_full = single.grads[off:off + cnt].numpy()
            single.adam(group, 1e-3, 0.5, 0.9, 1e-8, 1.0)
            if rank == 0:
                q.put((group + '_grad_err', float(np.abs(g_dp - g_full).max() / np.abs(g_full).max())))
                q.put((group + '_param_err', float((eng.params - single.params).abs().max())))
        t = eng.params.clone()
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        if rank == 0:
            q.put(('replica_diff', float((gathered[0] - gathered[1]).abs().max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gan_phases_equal_big_batch():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(7))
    for g in ('Discriminator', 'Generator', 'Encoder'):
        assert res[g + '_grad_err'] < 1e-10, res
        assert res[g + '_param_err'] < 1e-9, res
    assert res['replica_diff'] == 0.0


def _agree_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unsupervised_anomaly_detection_brain_mri_amd.parallel import _all_ok
        # a step that fails on ONE rank only must come out as a failure on EVERY rank (a rank on the torch path beside ranks
        # issuing ncclAllReduce on the library's communicator deadlocks the first step)
        a = _all_ok(True)
        b = _all_ok(rank != 1)
        c = _all_ok(rank != 0)
        d = _all_ok(False)
        q.put((rank, a, b, c, d))
    finally:
        dist.destroy_process_group()


def test_library_path_decision_is_collective():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, False, False, False), (1, True, False, False, False)], got
