"""Root-cause of the round-1 red test (tests/test_gpu_trainers.py::test_context_encoder_trainer[autoencoder_spatial]): the same
TRAIN step in both math modes against the fp64 oracle, every gradient tensor, run twice (determinism), with a count of the oracle's
pre-activations / L1 differences that lie within fp32 round-off of a kink.  Run on the GPU box: python tests/debug/ce_spatial_rootcause.py"""
import importlib
import os
import random
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae as ovae  # noqa: E402
from unsupervised_anomaly_detection_brain_mri_amd.trainers import CE  # noqa: E402
from unsupervised_anomaly_detection_brain_mri_amd.trainers.CE import retrieve_masked_batch  # noqa: E402
from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options  # noqa: E402
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    for mname in ('autoencoder_spatial', 'autoencoder'):
        net = getattr(importlib.import_module(f'unsupervised_anomaly_detection_brain_mri_amd.models.{mname}'), mname)
        opt = get_options(batchsize=4, learningrate=1e-3, numEpochs=3, zDim=64, outputWidth=64, outputHeight=64,
                          config={'CHECKPOINTDIR': tmp + '/ck', 'SAMPLEDIR': tmp + '/smp'})
        ds = SyntheticDataset(32, 16, 64, 64, seed=0)
        cfg = get_config(CE, opt, 'ADAM', [8, 8], 0.2, ds)
        model = CE(None, cfg, network=net)
        model.mask_rng = random.Random(0)
        x, _, bm = ds.next_batch(4, set='TRAIN', return_brainmask=True)
        x_ce = retrieve_masked_batch(x, bm)
        m = ovae.SpatialAE(64, 64, 1, 8) if mname == 'autoencoder_spatial' else ovae.Model('AE', 64, 64, 1, 8, 64)
        p = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
        masks = model._draw(4, True)[1]
        m64 = {k: v.astype(np.float64) for k, v in masks.items()}
        args = (p, x_ce.astype(np.float64), m64) if mname == 'autoencoder_spatial' else (p, x_ce.astype(np.float64), None, m64)
        out, cache = m.forward(*args)
        g = m.backward(p, x.astype(np.float64), out, cache, m64)
        print(f'==== {mname}')
        for k, v in cache.items():
            if 'bn' in k:
                a = np.abs(v)
                nz = a[a > 0]
                print(f'  kink census {k:12s} n={v.size:8d} exact0={int((a == 0).sum()):7d} '
                      f'|v|<1e-6:{int((nz < 1e-6).sum())} <1e-5:{int((nz < 1e-5).sum())} <1e-4:{int((nz < 1e-4).sum())} max={a.max():.3g}')
        d = np.abs(out['x_hat'] - x.astype(np.float64))
        nz = d[d > 0]
        print(f'  kink census L1 diff     n={d.size} exact0={int((d == 0).sum())} <1e-7:{int((nz < 1e-7).sum())} <1e-6:{int((nz < 1e-6).sum())} <1e-5:{int((nz < 1e-5).sum())}')
        res = {}
        for math in ('f32', 'bf16x3', 'f32', 'bf16x3'):
            model.engine.set_math(math)
            got = model.engine.forward(x, None, masks, want_backward=True, x_ce=x_ce)
            model.engine.backward()
            grads = model.engine.get_grads()
            xh = got['x_hat'].cpu().numpy()
            sg_dev = np.sign(xh.astype(np.float64) - x)
            sg_ref = np.sign(out['x_hat'] - x)
            print(f'  -- math {math}: x_hat rel {np.abs(xh - out["x_hat"]).max() / np.abs(out["x_hat"]).max():.2e}  L1-sign flips {int((sg_dev != sg_ref).sum())}')
            if math in res:
                same = all(np.array_equal(res[math][k], grads[k]) for k in grads)
                print(f'     second run bit-identical to the first: {same}')
                continue
            res[math] = grads
            for name in g:
                e = np.abs(grads[name] - g[name]).max() / max(np.abs(g[name]).max(), 1e-30)
                flag = '  <-- > 1e-4' if e > 1e-4 else ''
                print(f'     {name:44s} rel {e:.2e}{flag}')
        for name in g:
            e = np.abs(res['f32'][name].astype(np.float64) - res['bf16x3'][name]).max() / max(np.abs(g[name]).max(), 1e-30)
            print(f'     f32 vs bf16x3 {name:44s} rel {e:.2e}')
        model.engine.close()


if __name__ == '__main__':
    main()
