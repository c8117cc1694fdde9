"""trainers/VAE_You.py — VAE with gradient-based (MAP) restoration at test time (You et al.): trained exactly like trainers/VAE.py
(:40-48, 56-57); `reconstruct()` runs `restore_steps` iterations of  x -= restore_lr * d(rec_n + kl_n + tv_lambda * TV_n(x - x_hat))/dx
(:52-53, 133-144) -- here on the device, batched, with no host round trip per step (uad_restore_step on a VAE handle);
`determine_best_lambda()` (:153-173)."""
import numpy as np
import torch

from .AEMODEL import Phase
from .VAE import VAE


class VAE_You(VAE):
    ARCHS = ('VAE',)                  # restoration runs on the fused VAE handle (uad_restore_step)
    class Config(VAE.Config):
        def __init__(self):          # trainers/VAE_You.py:12-17
            super().__init__()
            self.modelname = 'VAE_You'
            self.restore_lr = 1e-3
            self.restore_steps = 150
            self.tv_lambda = 1.8

    def __init__(self, sess, config=None, network=None, **kw):
        super().__init__(sess, config, network, **kw)
        c = self.config
        self.restore_lr = float(getattr(c, 'restore_lr', 1e-3))
        self.restore_steps = int(getattr(c, 'restore_steps', 150))
        self.tv_lambda_value = float(getattr(c, 'tv_lambda', 1.8))

    def restore_gradients(self, x, eps=None, tv_lambda=None):
        """The `grads` fetch (:53) at x, without moving x."""
        xr = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.engine.device)
        if eps is None:
            eps = self.rng.standard_normal((len(x), self.config.zDim)).astype(np.float32)
        tv = self.tv_lambda_value if tv_lambda is None else tv_lambda
        return self.engine.restore_step(xr, None, eps, tv_lambda=tv, restore_lr=0.0, want_grads=True).cpu().numpy()

    def _restore(self, x, steps, tv_lambda, eps=None):
        """steps x (x -= restore_lr * grads) on device; fresh z noise per step like the graph's tf.random_normal (eps: optional callable
        step -> [n,zDim], or a fixed array; eps=0.0 pins the noise to zero)."""
        xr = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.engine.device)
        n, zd = len(x), self.config.zDim
        g = torch.Generator(device=self.engine.device).manual_seed(int(self.rng.integers(1 << 31)))
        for step in range(steps):
            if eps is None:
                e = torch.randn((n, zd), device=self.engine.device, generator=g)
            else:
                e = eps(step) if callable(eps) else eps
            self.engine.restore_step(xr, None, e, tv_lambda=tv_lambda, restore_lr=self.restore_lr)
        return xr.cpu().numpy()

    def reconstruct(self, x, dropout=False, eps=None):      # trainers/VAE_You.py:129-151
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        if eps is not None and np.isscalar(eps):
            eps = (lambda step: None) if float(eps) == 0.0 else None
        bs = self.engine.max_batch
        rec = np.concatenate([self._restore(x[s0:s0 + bs], self.restore_steps, self.tv_lambda_value, eps) for s0 in range(0, len(x), bs)], axis=0)
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}

    def determine_best_lambda(self, dataset):                # :153-173
        lambdas = np.arange(20) / 10.0
        mean_errors = []
        for tv_lambda in lambdas:
            errors = []
            for _ in range(int(dataset.num_batches(self.config.batchsize, set=Phase.VAL.value) * 0.2)):
                batch, _, _ = dataset.next_batch(self.config.batchsize, set=Phase.VAL.value)
                batch = batch.cpu().numpy() if hasattr(batch, 'cpu') else np.asarray(batch)
                errors.append(np.sum(np.abs(batch - self._restore(batch, self.restore_steps, float(tv_lambda)))))
            mean_error = np.mean(errors) if errors else np.nan
            mean_errors.append(mean_error)
            print(f'mean_error for lambda {tv_lambda}: {mean_error}')
        self.tv_lambda_value = lambdas[mean_errors.index(min(mean_errors))]
        print(f'Best lambda: {self.tv_lambda_value}')
