"""trainers/AEMODEL.py — Config defaults, model_dir, checkpoint resume, early stopping, plus the shared
process()/step()/reconstruct() machinery of the AE-family trainers (trainers/AE.py:63-110, VAE.py:76-123)."""
import os
from collections import defaultdict
from enum import Enum
from math import inf

import numpy as np

from .DLMODEL import DLMODEL
from ..engine import Engine
from ..parallel import DataParallelStep


class Phase(Enum):          # utils/logger.py:8-11
    TRAIN = 'TRAIN'
    VAL = 'VAL'
    TEST = 'TEST'


def indicate_early_stopping(current_cost, best_cost, last_improvement):   # trainers/AEMODEL.py:70-79
    if current_cost < best_cost:
        return current_cost, 0, False
    last_improvement += 1
    return best_cost, last_improvement, last_improvement >= 5


class AEMODEL(DLMODEL):
    class Config(DLMODEL.Config):
        def __init__(self, modelname='AE'):       # trainers/AEMODEL.py:13-23
            super().__init__()
            self.modelname = modelname
            self.intermediateResolutions = [8, 8]
            self.outputWidth = 256
            self.outputHeight = 256
            self.numChannels = 3
            self.dropout = False
            self.dropout_rate = 0.2
            self.zDim = 128

    ARCH = 'AE'
    SCALAR_KEYS = ('reconstructionLoss', 'loss')

    @property
    def ARCHS(self):                  # network families this trainer class accepts
        return (self.ARCH,)

    def __init__(self, sess, config=None, network=None, seed=0, world=None, device=None):
        super().__init__(sess, config if config is not None else self.Config())
        if network is None or not hasattr(network, 'arch'):
            raise ValueError('network= must be one of unsupervised_anomaly_detection_brain_mri_amd.models.*')
        if network.arch not in self.ARCHS:
            raise ValueError(f'trainer {type(self).__name__} expects a {" / ".join(self.ARCHS)} network, got {network.__name__}')
        self.arch = network.arch
        self.network = network
        self.losses = {}
        c = self.config
        self.checkpointDir = os.path.join(c.checkpointDir or 'checkpoints', self.network.__name__)
        self.engine = self._make_engine(device)
        self.dp = self._make_dp(world)
        if getattr(self.dp, 'world', 1) > 1 and hasattr(self.engine, 'set_fault_deferred'):
            # data parallel: a bottleneck fault is reported only by the agreement at the end of process() -- a rank raising alone out of a mid-epoch
            # forward would leave the others blocked in the next gradient all-reduce
            self.engine.set_fault_deferred(True)
        self.rng = np.random.default_rng(seed)       # host RNG: variable initialisation, and eps / masks when device_noise is off
        # eps / dropout masks of a step are drawn ON THE DEVICE by a counter-based generator keyed (seed, step, global sample index):
        # no host RNG, no H2D copy per step, and the same noise whichever rank holds the sample
        self.device_noise = True
        self.noise_seed = int(seed)
        self.noise_step = 0
        self.initialize_variables()
        self.get_number_of_trainable_params()

    def _make_engine(self, device):
        c = self.config
        return Engine(self.arch, c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]),
                      c.zDim, max_batch=max(int(c.batchsize), 1), device=device)

    def _make_dp(self, world):
        return DataParallelStep(self.engine, world)

    def initialize_variables(self):
        """tf.global_variables_initializer(): glorot_uniform kernels, zero bias, gamma 1, beta 0 (SURVEY §8a note 3)."""
        flat = np.zeros(self.engine.nparams, np.float32)
        rng = np.random.default_rng(int(self.rng.integers(1 << 31)))
        for name, shape, off in self.engine.spec:
            cnt = int(np.prod(shape))
            if name.endswith('kernel'):
                rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
                flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
            elif name.endswith('gamma'):
                flat[off:off + cnt] = 1.0
            elif name == 'Variable':      # gaussian_mixture_variational_autoencoder_spatial.py:43: tf.constant(0.1)
                flat[off:off + cnt] = 0.1
        self.engine.set_params(flat)
        self.engine.reset_optimizer()
        self.dp.broadcast_params(0)

    @property
    def model_dir(self):            # trainers/AEMODEL.py:54-61
        c = self.config
        return "{}_d{}_s{}x{}_{}_b{}_z{}_{}".format(c.modelname, c.dataset, c.outputWidth, c.outputHeight,
                                                    self.network.__name__, c.batchsize, c.zDim, c.description)

    def log_to_tensorboard(self, epoch, scalars, visuals, phase, name='x'):      # trainers/AEMODEL.py:37-42
        """Epoch means of the scalar fetches (+ up to 50 image rows when `visuals` is given) into <logDir>/{TRAIN,VAL,TEST} event files
        (utils/logger.py, TensorBoard's format, no TensorFlow).  The reference only summarises when visuals were collected; the scalars are
        written here in either case.  Off with config.useTensorboard = False."""
        if not getattr(self.config, 'useTensorboard', False):
            return
        if getattr(self, 'logger', None) is None:
            from ..utils.logger import Logger
            base = getattr(self.config, 'logDir', None) or os.path.join(os.path.dirname(os.path.abspath(self.checkpointDir)), 'logs')
            import time
            self.logDir = os.path.join(base, self.network.__name__, self.model_dir, time.strftime('%Y%m%d_%H%M%S'))      # AEMODEL.py:33-34
            self.logger = Logger(None, self.logDir)
        summ = {k: np.float32(np.mean(v)) for k, v in scalars.items()}
        if visuals:
            summ[name] = np.vstack(visuals)[:50]
        self.logger.summarize(epoch, phase=phase, summaries_dict=summ)

    def load_checkpoint(self):      # trainers/AEMODEL.py:44-52
        could_load, counter = self.load(self.checkpointDir)
        print(" [*] Load SUCCESS" if could_load else " [!] Load failed...")
        return counter if could_load else 0

    # ------------------------------------------------------------------ RNG inputs of one sess.run
    def _draw(self, n, dropout):
        """Host draw (numpy) of (eps, masks): tests that inject noise, and device_noise = False."""
        raise NotImplementedError

    def _noise_layout(self, dropout):
        """[(name, per-sample shape, 'normal' | 'keep')] of the arrays one sess.run draws ('eps' + the dropout masks by io name)."""
        raise NotImplementedError

    @property
    def rank(self):
        import torch.distributed as dist
        return dist.get_rank() if (self.dp.world > 1 and dist.is_initialized()) else 0

    def _noise(self, n, dropout):
        """(eps, masks) of the next sess.run.  Device path: one launch of the counter-based generator; sample i of this rank is global
        sample rank * n + i of step noise_step (contiguous partitioning of the global batch, parallel.py)."""
        if not self.device_noise:
            return self._draw(n, dropout)
        from ..engine import rng_fill
        layout = self._noise_layout(dropout)
        step = self.noise_step
        self.noise_step += 1
        if not layout:
            return None, None
        r = float(self.config.dropout_rate)
        got = rng_fill([(name, shape, kind, r) for name, shape, kind in layout], n, self.noise_seed, step, self.rank * n,
                       device=self.engine.device)
        eps = got.pop('eps', None)
        return eps, (got or None)

    # ------------------------------------------------------------------ one sess.run
    def _run(self, batch, phase, eps=None, dropout_masks=None, fetch_maps=True, **kw):
        """Enqueues one sess.run on the device and returns the engine's dict of DEVICE tensors (no host synchronisation)."""
        train = phase == Phase.TRAIN
        n = len(batch)
        if eps is None and dropout_masks is None:
            eps, masks = self._noise(n, dropout=train)
        else:
            d_eps, d_masks = self._draw(n, dropout=train)
            eps = d_eps if eps is None else eps
            masks = d_masks if dropout_masks is None else dropout_masks
        c = self.config
        kw.setdefault('want_latents', False)      # the reference's process() fetches no latent (VAE.py:83-96); reconstruct() does
        if train:
            return self.dp.train_step(batch, eps, masks, lr=c.learningrate, beta1=c.beta1, want_l1=fetch_maps, **kw)
        return self.engine.forward(batch, eps, masks, want_backward=False, want_l1=fetch_maps, **kw)

    def _scalars_to_run(self, sc):
        run = {'reconstructionLoss': np.float32(sc[0]), 'loss': np.float32(sc[2])}
        if 'kl' in self.SCALAR_KEYS:
            run['kl'] = np.float32(sc[1])
        else:
            run['loss'] = run['reconstructionLoss']
        return run

    def step(self, batch, phase, *, eps=None, dropout_masks=None, fetch_maps=True):
        """The body of the reference's process() loop (= one sess.run, VAE.py:83-96): returns the same fetch keys as host values.
        TRAIN runs fwd + bwd + Adam; VAL/TEST run the forward + losses only (dropout False).
        eps / dropout_masks may be injected (parity tests); otherwise they are drawn on the device (see _noise)."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        out = self._run(batch, phase, eps, dropout_masks, fetch_maps)
        sc = self.dp.allreduce_scalars(out['scalars'].clone()).cpu().numpy()       # the only host sync of the step
        run = self._scalars_to_run(sc)
        if fetch_maps:
            run['reconstruction'] = out['x_hat'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
        return run

    def _shard(self, dataset, phase, **kw):
        """This rank's slice of the next GLOBAL batch (config.batchsize slices per rank, contiguous partitioning): every rank advances the
        same dataset cursor over batchsize * world slices and keeps rows [rank * bs, (rank + 1) * bs)."""
        bs, w = self.config.batchsize, getattr(getattr(self, 'dp', None), 'world', 1)
        got = dataset.next_batch(bs * w, set=phase.value, **kw)
        if w == 1:
            return got
        lo = self.rank * bs
        return tuple(None if a is None else a[lo:lo + bs] for a in got)

    def _num_batches(self, dataset, phase):
        """Steps of one epoch: the dataset is walked in GLOBAL batches of config.batchsize * world slices (every rank takes its share, _shard)."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        world = getattr(getattr(self, 'dp', None), 'world', 1)
        nb = dataset.num_batches(self.config.batchsize * world, set=phase.value)
        if nb == 0:
            # (the reference's loop would run zero steps and then fail on the empty scalar dict: say what is wrong instead)
            raise ValueError(f'{phase.value} split holds fewer slices than one global batch (batchsize {self.config.batchsize} x {world} rank(s)): '
                             f'lower config.batchsize or the rank count')
        return nb

    def process(self, dataset, epoch, phase, optim=None):       # trainers/VAE.py:76-103
        """One epoch.  The loop body only ENQUEUES work: the batch comes from the dataset (device tensors when it is an HBM-resident
        utils.slice_cache.DeviceDataset), the noise is drawn on the device, and every step's scalar fetches are copied into one row of a
        device table; the table is read back once per epoch (the reference's per-step console line is printed then, same text).
        config.tfSummaryImages restores the reference's per-step map fetch for its TensorBoard image strip."""
        import torch
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        # Data parallel: for the duration of THIS epoch loop a bottleneck fault is only recorded on the device and reported by the cross-rank agreement
        # below -- a rank raising alone out of a mid-epoch forward would leave the others blocked in the next gradient all-reduce.  Outside the loop
        # (reconstruct(), evaluation, save()) the handle reports at once again.
        deferred = getattr(getattr(self, 'dp', None), 'world', 1) > 1 and hasattr(self.engine, 'set_fault_deferred')
        if deferred:
            self.engine.set_fault_deferred(True)
        try:
            return self._process_epoch(dataset, epoch, phase, optim)
        finally:
            if deferred and getattr(self.engine, 'handle', None):
                self.engine.set_fault_deferred(False)

    def _process_epoch(self, dataset, epoch, phase, optim):
        import torch
        if getattr(self.step, '__func__', None) is not AEMODEL.step:
            return self._process_via_step(dataset, epoch, phase)
        visuals = []
        # the reference fetches the maps of EVERY step for its TensorBoard image strip (trainer_utils.get_summary_dict); here that is opt-in
        # (config.tfSummaryImages): the maps are 4 MB of D2H per step
        want_images = bool(getattr(self.config, 'tfSummaryImages', False)) and bool(getattr(self.config, 'useTensorboard', False))
        num_batches = self._num_batches(dataset, phase)
        table = torch.zeros((max(num_batches, 1), 8), device=self.engine.device)
        for idx in range(num_batches):
            batch, _, _ = self._shard(dataset, phase)
            # the fused handle writes its 8 scalars straight into this step's table row; the materialised-graph engines return their own (fewer)
            kw = {'scalars_out': table[idx]} if getattr(self.engine, 'SCALARS_IN_PLACE', False) else {}
            out = self._run(batch, phase, fetch_maps=want_images, **kw)
            sc = out['scalars'].reshape(-1)
            if sc.data_ptr() != table[idx].data_ptr():
                table[idx, :sc.numel()].copy_(sc)
            if want_images:
                from .trainer_utils import get_summary_dict
                b = batch.cpu().numpy() if hasattr(batch, 'cpu') else np.asarray(batch)
                run = self._scalars_to_run(out['scalars'].cpu().numpy())
                run['reconstruction'], run['L1'] = out['x_hat'].cpu().numpy(), out['L1'].cpu().numpy()
                visuals.append(get_summary_dict(b, run)[1])
        # Fault word of the fused bottleneck (uad_check_fault): read after a stream sync and agreed on across ranks IN the epoch's one collective
        # (an extra table row), so that a rank whose kernels timed out does not raise alone while the others block in the next all-reduce.  Under
        # data parallelism the handle is in deferred-report mode (__init__): no forward of the loop above can have raised on one rank only, every
        # rank has issued every collective of the epoch, and all ranks raise here together.
        fault = None
        if hasattr(self.engine, 'check_fault'):
            try:
                self.engine.check_fault(sync=True)
            except RuntimeError as e:
                fault = e
        table = torch.cat([table, torch.full((1, 8), 1.0 if fault else 0.0, device=table.device)])
        allrows = self.dp.allreduce_scalars(table).cpu().numpy()                      # the epoch's one host synchronisation
        if fault is not None or allrows[-1, 0] > 0:
            raise fault if fault is not None else RuntimeError('another rank reported a fused-bottleneck fault in this epoch (its optimizer updates were '
                                                               'skipped on the device): replicas have diverged, restart from the last checkpoint')
        rows = allrows[:num_batches]
        scalars = defaultdict(list)
        quiet = bool(getattr(self.config, 'quiet', False))
        for idx, sc in enumerate(rows):
            run = self._scalars_to_run(sc)
            if not quiet:
                print(f'Epoch ({phase.value}): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["loss"]:.8f}')
            for k, v in run.items():
                scalars[k].append(v)
        out = {k: np.mean(v) for k, v in scalars.items()}
        for k, v in out.items():
            self.curves.setdefault(f'{phase.value}/{k}', []).append(float(v))
        self.log_to_tensorboard(epoch, out, visuals, phase)
        return out

    def _process_via_step(self, dataset, epoch, phase):
        """The epoch loop of trainers whose step() is their own (multi-phase GAN trainers): one step() -- with its host fetch -- per batch."""
        scalars = defaultdict(list)
        visuals = []
        want_images = bool(getattr(self.config, 'tfSummaryImages', False)) and bool(getattr(self.config, 'useTensorboard', False))
        num_batches = self._num_batches(dataset, phase)
        for idx in range(num_batches):
            batch, _, _ = self._shard(dataset, phase)
            run = self.step(batch, phase, fetch_maps=want_images)
            print(f'Epoch ({phase.value}): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["loss"]:.8f}')
            for k, v in run.items():
                if np.ndim(v) == 0:
                    scalars[k].append(v)
            if want_images:
                from .trainer_utils import get_summary_dict
                b = batch.cpu().numpy() if hasattr(batch, 'cpu') else np.asarray(batch)
                visuals.append(get_summary_dict(b, run)[1])
        out = {k: np.mean(v) for k, v in scalars.items()}
        for k, v in out.items():
            self.curves.setdefault(f'{phase.value}/{k}', []).append(float(v))
        self.log_to_tensorboard(epoch, out, visuals, phase)
        return out

    def train(self, dataset):       # trainers/VAE.py:31-74
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        best_cost, last_improvement = inf, 0
        # TRAIN must hold one global batch (batchsize x ranks): _num_batches raises the clear error NOW, not after the first epoch.  A VAL split that
        # is too small for one global batch -- the global batch grows with the rank count, so a split that validates on 1 GPU may not on 8 -- only
        # switches validation and early stopping off, as before round 4 (a resumed run keeps working); said once, on rank 0.
        self._num_batches(dataset, Phase.TRAIN)
        try:
            self._num_batches(dataset, Phase.VAL)
            have_val = True
        except ValueError as e:
            have_val = False
            if self.rank == 0:
                print(f'warning: {e}; training without validation / early stopping')
        last_epoch = self.load_checkpoint()
        for epoch in range(last_epoch, self.config.numEpochs):
            self.process(dataset, epoch, Phase.TRAIN, optim=True)
            last_epoch += 1
            if self.rank == 0:                      # replicas are identical: one writer
                self.save(self.checkpointDir, last_epoch)
            if not have_val:
                continue
            val_scalars = self.process(dataset, epoch, Phase.VAL)
            best_cost, last_improvement, stop = indicate_early_stopping(val_scalars['loss'], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break

    def reconstruct(self, x, dropout=False, eps=None):
        """trainers/VAE.py:105-123: {'reconstruction', 'l1err', 'l2err'} (l2err == l1err, sic).  The reference samples
        z = mu + eps*sigma at eval too: eps is drawn unless given; pass eps=0 for the deterministic mode."""
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        d_eps, masks = self._noise(len(x), dropout=bool(dropout))
        if eps is None:
            eps = d_eps
        elif np.isscalar(eps):
            eps = None if float(eps) == 0.0 else np.full((len(x), self.config.zDim), eps, np.float32)
        out = self.engine.forward(x, eps, masks, want_backward=False, want_l1=False, want_latents=False)
        rec = out['x_hat'].cpu().numpy()
        results = {'reconstruction': rec}
        results['l1err'] = np.sum(np.abs(x - rec))
        results['l2err'] = np.sum(np.sqrt((x - rec) ** 2))
        return results
