import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


# Collection order of the -m gpu run (the driver runs it with -x): SURVEY.md section 8(a)-(e) hot-path rows first, the section 8(f)
# widening (model variants, dataset cache, checkpoint interop) last, so that a failure in a variant cannot hide the hot-path rows.
_ORDER = ['test_gpu_ops.py', 'test_gpu_ops_large.py', 'test_gpu_model.py', 'test_gpu_scale_parity.py', 'test_gpu_shapes.py',
          'test_gpu_optimizers.py', 'test_gpu_cevae.py', 'test_gpu_gmvae.py', 'test_gpu_fanogan.py', 'test_gpu_ops_resnet.py',
          'test_gpu_eval.py', 'test_gpu_trainers.py', 'test_gpu_rng.py', 'test_gpu_dp_rehearsal.py']
_VARIANT_TESTS = ('context_encoder', 'latent_ae', 'anovaegan', 'aae_family', 'gmvae_dense', 'tf_checkpoint')


def pytest_collection_modifyitems(config, items):
    def rank(item):
        f = os.path.basename(str(item.fspath))
        r = _ORDER.index(f) if f in _ORDER else len(_ORDER) + (1 if f.startswith('test_gpu') else 0)
        if any(v in item.name for v in _VARIANT_TESTS):
            r = len(_ORDER) + 1
        return r
    items.sort(key=rank)          # stable: file / definition order is kept inside a rank
