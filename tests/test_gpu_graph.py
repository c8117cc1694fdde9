"""GPU: hipGraph replay of whole phases (uad_gan_set_graph_mode) is bit-identical to the plain launch path -- same kernels in the same
order -- for the f-AnoGAN phases, the AAE-family phases and the dense GMVAE's train / restore steps, over several optimisation steps
(so that first sight, capture and replay of every key are all exercised), and falls back cleanly when keys churn."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
except Exception:
    GanEngine = None


def _init(eng, seed=3):
    rng = np.random.default_rng(seed)
    flat = np.zeros(eng.nparams, np.float32)
    for name, shape, off in eng.spec:
        cnt = int(np.prod(shape))
        if name.endswith('kernel'):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
        elif name.endswith('gamma'):
            flat[off:off + cnt] = 1.0
        elif name == 'Variable':
            flat[off:off + cnt] = 0.1
    eng.set_params(flat)


def _pair(**kw):
    a, b = GanEngine(graph=False, **kw), GanEngine(graph=True, **kw)
    _init(a); _init(b)
    return a, b


def test_fanogan_phases_graph_equals_plain():
    n, h, zd = 4, 32, 16
    plain, graphed = _pair(height=h, width=h, inter_res=8, zdim=zd, max_batch=n, variant='unified')
    rng = np.random.default_rng(0)
    for it in range(5):
        x = synthetic_slices(n, h, h, seed=it)
        z = rng.standard_normal((n, zd)).astype(np.float32)
        alpha = rng.random(n).astype(np.float32)
        for eng in (plain, graphed):
            eng.res = []
            for group, kw in (('Generator', dict(z=z)), ('Discriminator', dict(x=x, z=z, alpha=alpha)), ('Encoder', dict(x=x))):
                out = eng.phase(group, want_l1=group == 'Encoder', **kw)
                eng.res.append({k: v.clone() for k, v in out.items()})
                eng.adam(group, 1e-4)
        for ra, rb in zip(plain.res, graphed.res):
            assert set(ra) == set(rb)
            for k in ra:
                assert torch.equal(ra[k], rb[k]), (it, k)
    assert np.array_equal(plain.get_buffer_host(_lib.BUF_PARAMS), graphed.get_buffer_host(_lib.BUF_PARAMS))
    st = graphed.graph_stats()
    assert st['enabled'] and st['captures'] >= 3 and st['replays'] >= 6, st
    assert plain.graph_stats()['captures'] == 0
    r0, r1 = plain.reconstruct(x), graphed.reconstruct(x)
    assert torch.equal(r0['reconstruction'], r1['reconstruction'])
    plain.close(); graphed.close()


def test_aae_and_gmvae_graph_equals_plain():
    n, h = 4, 32
    plain, graphed = _pair(height=h, width=h, inter_res=8, zdim=16, max_batch=n, variant='aae', aae_kind='constrained_aae')
    rng = np.random.default_rng(1)
    for it in range(4):
        x = synthetic_slices(n, h, h, seed=10 + it)
        z = rng.standard_normal((n, 16)).astype(np.float32)
        eps = rng.random(n).astype(np.float32)
        mz = ((rng.random((n, 16)) >= 0.2) / 0.8).astype(np.float32)
        outs = []
        for eng in (plain, graphed):
            cl = lambda d: {k: v.clone() for k, v in d.items()}       # graph mode: results live in handle-owned buffers until the next call
            o = [cl(eng.aae_phase('AE', x, mask_z=mz, want_l1=True))]
            eng.adam('AE', 1e-4)
            o.append(cl(eng.aae_phase('Discriminator', x, z=z, eps=eps, mask_z=mz))); eng.adam('Discriminator', 1e-4)
            o.append(cl(eng.aae_phase('Encoder', x, mask_z=mz))); eng.adam('Encoder', 1e-4)
            outs.append(o)
        for da, db in zip(*outs):
            for k in da:
                assert torch.equal(da[k], db[k]), (it, k)
    assert np.array_equal(plain.get_buffer_host(_lib.BUF_PARAMS), graphed.get_buffer_host(_lib.BUF_PARAMS))
    assert graphed.graph_stats()['replays'] >= 3
    plain.close(); graphed.close()

    plain, graphed = _pair(height=h, width=h, inter_res=8, zdim=2, max_batch=n, variant='aae', aae_kind='gmvae', dim=5, dim_w=3)
    x = synthetic_slices(n, h, h, seed=3)
    xs = [torch.from_numpy(x.copy()).cuda(), torch.from_numpy(x.copy()).cuda()]
    for it in range(5):
        e_w, e_z = rng.standard_normal((n, 3)).astype(np.float32), rng.standard_normal((n, 2)).astype(np.float32)
        res = []
        for eng, xr in zip((plain, graphed), xs):
            o = {k: v.clone() for k, v in eng.gm_phase(x, e_w, e_z).items()}
            eng.adam('AE', 1e-4, 0.5, 0.999)
            g = eng.gm_restore_step(xr, e_w, e_z, tv_lambda=1.8, restore_lr=1e-2, want_grads=True)
            res.append((o, g.clone()))
        for k in res[0][0]:
            assert torch.equal(res[0][0][k], res[1][0][k]), (it, k)
        assert torch.equal(res[0][1], res[1][1])
    assert torch.equal(xs[0], xs[1])
    st = graphed.graph_stats()
    assert st['captures'] >= 2 and st['replays'] >= 4, st
    plain.close(); graphed.close()


def test_graph_cache_eviction_and_toggle():
    n, h = 2, 32
    eng = GanEngine(height=h, width=h, inter_res=8, zdim=16, max_batch=8, variant='aae', aae_kind='constrained_ae', graph=True)
    _init(eng)
    ref = GanEngine(height=h, width=h, inter_res=8, zdim=16, max_batch=8, variant='aae', aae_kind='constrained_ae', graph=False)
    _init(ref)
    # many distinct keys (batch sizes x flags), each seen three times: every one is captured, old ones are evicted, results stay right
    for rep in range(3):
        for nb in range(1, 9):
            x = synthetic_slices(nb, h, h, seed=nb)
            for wb in (False, True):
                a = eng.aae_phase('AE', x, want_backward=wb, want_l1=True)
                a = {k: v.clone() for k, v in a.items()}
                b = ref.aae_phase('AE', x, want_backward=wb, want_l1=True)
                assert torch.equal(a['reconstruction'], b['reconstruction']) and torch.equal(a['loss'], b['loss'])
    assert eng.graph_stats()['captures'] >= 16
    eng.set_graph_mode(False)
    x = synthetic_slices(n, h, h, seed=1)
    a, b = eng.aae_phase('AE', x), ref.aae_phase('AE', x)
    assert torch.equal(a['loss'], b['loss']) and not eng.graph_stats()['enabled']
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), ref.get_buffer_host(_lib.BUF_GRADS))
    eng.close(); ref.close()
