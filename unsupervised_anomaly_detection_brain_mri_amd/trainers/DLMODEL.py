"""trainers/DLMODEL.py — base class: Config, save/load, optimizer validation.  The TF session / Saver are replaced by
the HIP engine handle and a flat-fp32 .npz checkpoint (TF tensor-bundle checkpoints are read / exported via utils/tf_checkpoint.py;
the directory layout, file names and resume-by-epoch behaviour are kept: DLMODEL.py:63-110)."""
import json
import os
import re

import numpy as np

from .. import _lib

OPTIMIZERS = ('ADAM', 'SGD', 'MOMENTUM', 'RMS')   # DLMODEL.py:113-123


class DLMODEL(object):
    class Config(object):
        def __init__(self):          # trainers/DLMODEL.py:13-26
            self.modelname = ''
            self.model_config = {}
            self.checkpointDir = None
            self.description = ''
            self.batchsize = 6
            self.useTensorboard = True
            self.tensorboardPort = 8008
            self.useMatplotlib = False
            self.debugGradients = False
            self.tfSummaryAfter = 100
            self.dataset = ''
            self.beta1 = 0.5

    def __init__(self, sess, config=None):
        self.sess = sess                 # accepted for signature compatibility; unused (no tf.Session here)
        self.config = config if config is not None else self.Config()
        self.variables = {}
        self.curves = {}
        self.losses = None
        self.engine = None

    @property
    def model_dir(self):
        return "{}_d{}_b{}_{}".format(self.config.modelname, self.config.dataset, self.config.batchsize, self.config.description)

    def create_optimizer(self=None, type='ADAM', momentum=0.9):
        """DLMODEL.create_optimizer (:112-123): validates the optimizer string and selects the engine's update rule.  ADAM everywhere; SGD /
        MOMENTUM / RMS (TF-1.15 rules) on the fused AE-family handle, which is where the reference's trainers pass config.optimizer through."""
        if isinstance(self, str):                    # called as the reference's staticmethod: create_optimizer('ADAM')
            self, type = None, self
        if type not in OPTIMIZERS:
            raise ValueError('Invalid optimizer type')
        eng = getattr(self, 'engine', None)
        if type != 'ADAM':
            if eng is None or not hasattr(eng, 'set_optimizer'):
                raise NotImplementedError(f"optimizer {type!r}: implemented on the fused AE-family handle only (this trainer's handle applies ADAM)")
            eng.set_optimizer(type, momentum)
        elif eng is not None and hasattr(eng, 'set_optimizer'):
            eng.set_optimizer('ADAM')
        return type

    # Adam step counter(s) stored with a checkpoint (one per optimiser)
    def _adam_steps(self):
        return np.int64(self.engine.step_count)

    def _set_adam_steps(self, t):
        self.engine.step_count = int(t)

    def save(self, checkpoint_dir, step):
        try:                                    # data-parallel replicas hold identical weights: one writer
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
                return
        except ImportError:
            pass
        model_name = self.config.modelname + ".model"
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        os.makedirs(checkpoint_dir, exist_ok=True)
        eng = self.engine
        np.savez(os.path.join(checkpoint_dir, f'{model_name}-{step}.npz'),
                 params=eng.get_buffer_host(_lib.BUF_PARAMS), adam_m=eng.get_buffer_host(_lib.BUF_ADAM_M),
                 adam_v=eng.get_buffer_host(_lib.BUF_ADAM_V), adam_t=self._adam_steps(),
                 noise_step=np.int64(getattr(self, 'noise_step', 0)),        # counter of the device noise generator: a resumed run continues the eps / mask sequence
                 names=np.array([n for n, _, _ in eng.spec]))
        with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
            f.write(f'model_checkpoint_path: "{model_name}-{step}"\n')
        with open(os.path.join(checkpoint_dir, 'Config-{}.json'.format(step)), 'w') as outfile:
            try:
                json.dump({k: v for k, v in self.config.__dict__.items() if k != 'options'}, outfile, default=str)
            except Exception:
                print("Failed to save config json")
        np.save(os.path.join(checkpoint_dir, 'Curves.npy'), self.curves)

    def load(self, checkpoint_dir, iteration=None):
        print(" [*] Reading checkpoints...")
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        curves_file = os.path.join(checkpoint_dir, 'Curves.npy')
        if os.path.isfile(curves_file):
            self.curves = np.load(curves_file, allow_pickle=True).item()
        name = None
        if iteration is not None:
            name = self.config.modelname + '.model-' + str(iteration)
        else:
            ck = os.path.join(checkpoint_dir, 'checkpoint')
            if os.path.isfile(ck):
                mt = re.search(r'model_checkpoint_path: "(.*)"', open(ck).read())
                name = mt.group(1) if mt else None
        if name and os.path.isfile(os.path.join(checkpoint_dir, name + '.npz')):
            z = np.load(os.path.join(checkpoint_dir, name + '.npz'))
            self.engine.set_params(z['params'])
            self.engine.set_buffer_host(_lib.BUF_ADAM_M, z['adam_m'])
            self.engine.set_buffer_host(_lib.BUF_ADAM_V, z['adam_v'])
            self._set_adam_steps(z['adam_t'])
            if hasattr(self, 'noise_step'):
                # older checkpoints carry no counter: the optimizer's step count is the number of TRAIN draws so far
                self.noise_step = int(z['noise_step']) if 'noise_step' in z.files else int(np.max(z['adam_t']))
            counter = int(next(re.finditer(r'(\d+)(?!.*\d)', name)).group(0))
            print(" [*] Success to read {}".format(name))
            return True, counter
        if name and os.path.isfile(os.path.join(checkpoint_dir, name + '.index')):
            # a TensorFlow V2 checkpoint written by the reference's tf.train.Saver (DLMODEL.py:63-83): same variable names as the spec
            from ..utils import tf_checkpoint
            got = tf_checkpoint.bundle_to_flat(self.engine.spec, tf_checkpoint.read_checkpoint(os.path.join(checkpoint_dir, name)),
                                               beta1=getattr(self.config, 'beta1', 0.5))
            if got['missing']:
                raise ValueError(f"TF checkpoint {name} lacks variables of this model: {got['missing'][:4]} ...")
            self.engine.set_params(got['params'])
            if got['adam_m'] is not None:
                self.engine.set_buffer_host(_lib.BUF_ADAM_M, got['adam_m'])
                self.engine.set_buffer_host(_lib.BUF_ADAM_V, got['adam_v'])
                if got['adam_t'] is not None:
                    self._set_adam_steps(np.full(np.shape(self._adam_steps()), got['adam_t'], np.int64))
            counter = int(next(re.finditer(r'(\d+)(?!.*\d)', name)).group(0))
            print(" [*] Success to read TF checkpoint {}".format(name))
            return True, counter
        print(" [*] Failed to find a checkpoint")
        return False, 0

    def save_tf(self, checkpoint_dir, step):
        """Exports the model as a TensorFlow V2 tensor bundle (<modelname>.model-<step>.index / .data-00000-of-00001) under the
        reference's variable names, for tf.train.Saver().restore on the reference side."""
        from ..utils import tf_checkpoint
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        os.makedirs(checkpoint_dir, exist_ok=True)
        eng = self.engine
        t = int(np.atleast_1d(self._adam_steps())[0])
        prefix = os.path.join(checkpoint_dir, f'{self.config.modelname}.model-{step}')
        tf_checkpoint.write_checkpoint(prefix, tf_checkpoint.flat_to_bundle(
            eng.spec, eng.get_buffer_host(_lib.BUF_PARAMS), eng.get_buffer_host(_lib.BUF_ADAM_M), eng.get_buffer_host(_lib.BUF_ADAM_V),
            t, beta1=getattr(self.config, 'beta1', 0.5), beta2=getattr(self.config, 'beta2', 0.999)))
        return prefix

    def get_number_of_trainable_params(self):
        scopes = {}
        for name, shape, _ in self.engine.spec:
            scopes[name.split('/')[0]] = scopes.get(name.split('/')[0], 0) + int(np.prod(shape))
        for scope, cnt in scopes.items():
            print(f'#Params in {scope}: {cnt}')
        print(f'#Params in total: {self.engine.nparams}')
        return self.engine.nparams
