"""CPU (-m "not gpu"): host-side logic — the C-ABI library loads and exports every symbol include/uad_hip.h declares
(no compute without a GPU), the product Metrics match the reference-generated golden vectors, config plumbing, and
the loud failure when no GPU / no library is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from unsupervised_anomaly_detection_brain_mri_amd import _lib
from unsupervised_anomaly_detection_brain_mri_amd.trainers import Metrics
from unsupervised_anomaly_detection_brain_mri_amd.utils import default_config_setup as cfgs
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset, synthetic_slices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'scoring_golden.npz'))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'uad_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(uad_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/uad_hip.h but not exported by libuad_hip.so'
    assert declared == set(_lib.SYMBOLS), 'ctypes table and header disagree'
    assert b'gfx950' in _lib.load().uad_version()


def test_library_exports_nothing_but_the_header():
    """Built with -fvisibility=hidden (build.py): the dynamic symbol table defines exactly the functions include/uad_hip.h declares -- none of
    the C++ launch layer (uad_launch_*, uad_fail, ...) leaks out of the boundary."""
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if not os.path.exists(nm):
        pytest.skip('no nm in this image')
    out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in 'TtWwBbDdRrVv'}
    exported = {s for s in exported if not s.startswith(('__hip_', '_init', '_fini', '__bss', '_edata', '_end'))}
    header = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'uad_hip.h')).read(), flags=re.S)
    declared = set(re.findall(r'\b(uad_[a-z0-9_]+)\s*\(', header))
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob and b'conv_gemm_kernel' in blob


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Engine('VAE', 32, 32, 1, 8, 16, max_batch=2)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_product_metrics_match_reference_golden():
    pred, gt = G['med'].flatten(), G['lab'].astype(bool).flatten()
    assert abs(Metrics.compute_prc(pred, gt)[0] - float(G['auprc'])) < 1e-12
    assert abs(Metrics.compute_roc(pred, gt)[0] - float(G['auroc'])) < 1e-12
    lab = G['lab'].astype(np.int64).flatten()
    scores, threshs = Metrics.compute_dice_score(pred, lab, 5)
    np.testing.assert_allclose(threshs, G['dice_threshs'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(scores, G['dice_scores'], rtol=0, atol=1e-12)
    bs, bt = Metrics.compute_dice_curve_recursive(pred, lab, granularity=5)
    assert abs(bs - float(G['best_score'])) < 1e-12 and abs(bt - float(G['best_thr'])) < 1e-15
    for t, ref in zip((0.05, 0.1, 0.2), G['dice_at']):
        d = Metrics.dice(np.where(pred > t, 1, 0), lab)
        assert (np.isnan(d) and np.isnan(ref)) or abs(d - ref) < 1e-12


def test_options_and_config_keys_match_reference():
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.VAE import VAE
    opt = cfgs.get_options(batchsize=8, learningrate=1e-4, numEpochs=1, zDim=128, outputWidth=128, outputHeight=128)
    for k in ('sliceStart', 'sliceEnd', 'threshold', 'keepOnlyPositiveResiduals', 'applyHyperIntensityPrior',
              'medianFiltering', 'erodeBrainmask', 'numMonteCarloSamples'):
        assert k in opt
    assert opt['threshold'] == 'bestdice' and opt['sliceStart'] == 20 and opt['sliceEnd'] == 130
    ds = SyntheticDataset(16, 8, 32, 32)
    c = cfgs.get_config(VAE, opt, 'ADAM', [8, 8], 0.2, ds)
    assert c.modelname == 'VAE' and c.beta1 == 0.5 and c.numChannels == 1 and c.dataset == 'SyntheticDataset'
    with pytest.raises(ValueError):
        cfgs.get_datasets(opt, dataset='nope')


def test_optimizer_string_validation():
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.DLMODEL import DLMODEL
    assert DLMODEL.create_optimizer('ADAM') == 'ADAM'
    with pytest.raises(ValueError, match='Invalid optimizer type'):
        DLMODEL.create_optimizer('RMSProp')          # reference accepts 'RMS', not 'RMSProp' (A16)
    with pytest.raises(NotImplementedError):
        DLMODEL.create_optimizer('SGD')


def test_synthetic_dataset_duck_type():
    ds = SyntheticDataset(20, 8, 32, 32)
    assert ds.num_batches(8, set='TRAIN') == 2 and ds.num_channels == 1
    b, l, m = ds.next_batch(8, set='TRAIN', return_brainmask=True)
    assert b.shape == (8, 32, 32, 1) and b.dtype == np.float32 and 0 <= b.min() and b.max() <= 1 and m.shape == (8, 32, 32)
    x = synthetic_slices(4, 64, 64, seed=1)
    frac = (x > 0).mean()
    assert 0.3 < frac < 0.6 and x[:, 0, 0, 0].max() == 0.0


def test_retrieve_masked_batch_matches_reference_golden():
    """tests/golden/cemask_golden.npz = outputs of the reference's trainers/CE.py:retrieve_masked_batch on seeded
    inputs (made by tests/golden/make_cemask_golden.py).  The oracle restatement and the product's trainers/CE.py must
    both replay them exactly, 32x32 'box too small for a 20x20 square' case included."""
    import random
    from oracle import vae as ovae
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.CE import retrieve_masked_batch
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'cemask_golden.npz'))
    for i in range(4):
        bm, seed, holes = g[f'bm{i}'], int(g[f'seed{i}']), g[f'holes{i}']
        x = np.ones(bm.shape)
        a = ovae.retrieve_masked_batch(x, bm, random.Random(seed))
        b = retrieve_masked_batch(x.astype(np.float32), bm, random.Random(seed))
        assert np.array_equal(a == 0, holes), i
        assert np.array_equal(b == 0, holes), i


def test_evaluation_host_helpers_against_golden_and_oracle():
    """Host-side helpers of utils/Evaluation.py: squash_intensities / apply_brainmask against the reference-generated golden,
    the connected-component filter against the oracle restatement, lesion-wise detection counts on a constructed case."""
    from oracle import scoring as osc
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation as E
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scoring_golden.npz'))
    np.testing.assert_allclose(E.squash_intensities(g['diffs'][3]), g['squashed'], rtol=1e-12)
    for s in (0, 5, 11):
        assert np.array_equal(E.apply_brainmask(np.ones((64, 64)), g['bm'][s], erode=True).astype(np.uint8), g['eroded'][s])
    rng = np.random.default_rng(0)
    v = (rng.random((12, 32, 32)) < 0.04) * rng.random((12, 32, 32))
    assert np.array_equal(E.filter_3d_connected_components(v.copy()), osc.filter_3d_connected_components(v.copy()))
    v4 = v.reshape(3, 4, 32, 32)
    assert E.filter_3d_connected_components(v4.copy()).shape == v4.shape
    gt = np.zeros((25, 32, 32), int); gt[3:6, 5:9, 5:9] = 1; gt[22:24, 20:23, 20:23] = 1
    pr = np.zeros_like(gt); pr[4:7, 6:10, 6:10] = 1; pr[10:13, 1:4, 1:4] = 1; pr[0, 0, 0] = 1
    assert E.compute_detection_rate(pr, gt) == (1, 1, 1)       # hit lesion, 27-voxel false blob (the 1-voxel one is ignored), missed lesion
    x = np.full((8, 8), 0.8); xr = np.full((8, 8), 0.5); x[0, 0] = 0.3
    d = E.postprocess_slice(x, xr)
    assert d[0, 0] == 0 and np.allclose(d[1:, 1:], 0.3)


# ------------------------------------------------------------------ slice cache
def test_slice_cache_roundtrip_and_cursor(tmp_path):
    """Cache files are plain arrays + JSON; BatchCursor restates the cursor / epoch-wrap / shuffle arithmetic of
    dataloaders/BRAINWEB.py:411-457 (checked here against a direct restatement on image arrays, same permutation draws)."""
    from unsupervised_anomaly_detection_brain_mri_amd.utils import slice_cache as sc
    rng = np.random.default_rng(0)
    imgs = rng.random((23, 8, 8, 1)).astype(np.float32)
    labs = rng.integers(0, 11, (23, 8, 8)).astype(np.uint8)
    sets = np.array([0] * 13 + [1] * 6 + [2] * 4)
    rng.shuffle(sets)
    sc.write_cache(str(tmp_path / 'c'), imgs, sets, labs, patients=['p0', 'p1'], options={'sliceStart': 20})
    im2, lb2, index = sc.read_cache(str(tmp_path / 'c'))
    assert np.array_equal(np.asarray(im2), imgs) and np.array_equal(np.asarray(lb2), labs)
    assert index['sets'] == sets.tolist() and index['patients'] == ['p0', 'p1'] and index['options'] == {'sliceStart': 20}
    assert os.path.getsize(tmp_path / 'c' / 'slices.f32') == imgs.size * 4
    lut = sc.brainmask_lut()
    assert [int(lut[v]) for v in range(11)] == [0, 1, 1, 1, 0, 0, 0, 0, 1, 0, 1]          # BRAINWEB.py:466-476

    # reference arithmetic on arrays (the images of one split), same RNG stream
    def ref_batches(data, bs, nb, seed):
        r = np.random.default_rng(seed)
        cur, start, out = data.copy(), 0, []
        for _ in range(nb):
            if start + bs > len(cur):
                rest = cur[start:]
                cur = cur[r.permutation(len(cur))]
                start = bs - len(rest)
                out.append(np.concatenate([rest, cur[:start]]))
            else:
                out.append(cur[start:start + bs]); start += bs
        return out

    data = np.arange(13)
    cur = sc.BatchCursor(13, np.random.default_rng(5))
    got = [data[cur.next(4)] for _ in range(11)]
    for a, b in zip(got, ref_batches(data, 4, 11, 5)):
        assert np.array_equal(a, b)
    assert np.array_equal(np.concatenate(got[:3]), np.arange(12))        # first epoch is never shuffled (BRAINWEB.py:419 quirk)
    assert cur.epochs_completed == 3
    for b in got:                                                         # every wrapped batch is complete and in range
        assert len(b) == 4 and b.min() >= 0 and b.max() < 13
    # no shuffle: plain wrap-around
    c2 = sc.BatchCursor(5, np.random.default_rng(1))
    assert [c2.next(2, shuffle=False).tolist() for _ in range(4)] == [[0, 1], [2, 3], [4, 0], [1, 2]]


def test_trainer_utils_summary_dict():
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import trainer_utils as tu
    rng = np.random.default_rng(0)
    batch = rng.random((3, 8, 8, 1)).astype(np.float32)
    run = {'reconstruction': rng.random((3, 8, 8, 1)).astype(np.float32), 'L1': rng.random((3, 8, 8, 1)).astype(np.float32),
           'loss': np.float32(2.0), 'kl': np.float32(0.5), 'optimizer': None, 'nan': float('nan')}
    scalars, visuals = tu.get_summary_dict(batch, run)
    assert set(scalars) == {'loss', 'kl'} and visuals.shape == (3, 8, 24, 1)
    assert visuals.min() == 0.0 and visuals.max() == 255.0
    np.testing.assert_allclose(visuals[1, :, :8, 0], 255 * (batch[1, ..., 0] - batch[1].min()) / (batch[1].max() - batch[1].min()), rtol=1e-5)
    assert tu.normalize(np.full((4, 4), 3.0)).max() == 0.0                   # constant image -> 0 (cv2.NORM_MINMAX)
    _, v2 = tu.get_summary_dict(batch, run, ['L1'], batch)
    assert v2.shape == (3, 8, 24, 1)


def test_mains_pairings_resolve():
    """every mains/main_*.py names a trainer / model pair that exists and that the trainer accepts (the reference's 17 entry points)."""
    import glob
    import importlib
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, 'mains', 'main_*.py')))
    assert len(files) == 17
    pkg = 'unsupervised_anomaly_detection_brain_mri_amd'
    for f in files:
        src = open(f).read()
        m = re.search(r"set_defaults\(trainer='(\w+)', model='(\w+)'", src)
        assert m, f
        T = getattr(importlib.import_module(f'{pkg}.trainers.{m.group(1)}'), m.group(1))
        net = getattr(importlib.import_module(f'{pkg}.models.{m.group(2)}'), m.group(2))
        archs = T.ARCHS if isinstance(T.ARCHS, tuple) else (T.ARCH,)
        assert net.arch in archs, (os.path.basename(f), net.arch, archs)
        assert net.__name__ == m.group(2)


def test_trainer_rank_sharding_of_the_global_batch():
    """AEMODEL._shard: under slice-batch DP every rank advances the SAME dataset cursor over batchsize * world slices and keeps its
    contiguous share (the partitioning parallel.py's big-batch equivalence and the rank-invariant noise assume)."""
    import types
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, Phase
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    got = []
    for rank in range(2):
        t = object.__new__(VAE)
        t.config = types.SimpleNamespace(batchsize=4)
        t.dp = types.SimpleNamespace(world=2)
        type(t).rank = property(lambda self, r=rank: r)
        ds = SyntheticDataset(16, 8, 16, 16, seed=0)
        b, _, m = t._shard(ds, Phase.TRAIN, return_brainmask=True)
        assert b.shape == (4, 16, 16, 1) and m.shape == (4, 16, 16)
        got.append(b)
        del type(t).rank
    ref = SyntheticDataset(16, 8, 16, 16, seed=0).next_batch(8, set='TRAIN')[0]
    assert np.array_equal(np.concatenate(got), ref)


def test_batch_cursor_reproduces_the_reference_next_batch():
    """utils/slice_cache.BatchCursor against what the REFERENCE's BRAINWEB.next_batch returned on the same split vectors and seeds
    (tests/golden/cursor_golden.npz, written by tests/golden/make_cursor_golden.py from dataloaders/BRAINWEB.py:406-478 run in this container):
    no shuffle before the first epoch (:419 compares a dict with 0), the epoch wrap takes the rest of the old arrangement + the head of the
    freshly permuted one, the permutation acts on the CURRENT arrangement, TRAIN and VAL share one random stream; brain-mask label mapping."""
    from unsupervised_anomaly_detection_brain_mri_amd.utils import slice_cache as sc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cursor_golden.npz'))

    class GlobalShuffle:            # numpy.random.shuffle(arange(n)) on the global stream, as the reference draws its permutations
        def permutation(self, n):
            p = np.arange(n)
            np.random.shuffle(p)
            return p

    for case in range(4):
        sets = g[f'sets{case}']
        bs, calls, seed, shuffle, nb_train, nb_val = (int(v) for v in g[f'cfg{case}'])
        idx = {k: np.where(sets == k)[0] for k in (0, 1)}
        assert len(idx[0]) // bs == nb_train and len(idx[1]) // 2 == nb_val
        np.random.seed(seed)
        rng = GlobalShuffle()
        cur = {k: sc.BatchCursor(len(idx[k]), rng) for k in (0, 1)}
        vi = 0
        for c in range(calls):
            got = idx[0][cur[0].next(bs, bool(shuffle))]
            assert np.array_equal(got, g[f'train{case}'][c]), (case, c)
            if len(idx[1]) and c % 2 == 0:
                assert np.array_equal(idx[1][cur[1].next(2, bool(shuffle))], g[f'val{case}'][vi]), (case, c)
                vi += 1
        lut = sc.brainmask_lut()
        assert np.array_equal(lut[g[f'bm_labels{case}'].astype(np.uint8)], g[f'bm{case}'].astype(np.uint8))


def test_epoch_without_one_global_batch_is_a_clear_error():
    """trainers/AEMODEL._num_batches: a split smaller than batchsize x world used to run zero steps and fail later with KeyError('loss')."""
    import types
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.AEMODEL import AEMODEL, Phase

    class Split:
        def num_batches(self, bs, set):
            return 0 if bs > 64 else 3

    tr = types.SimpleNamespace(config=types.SimpleNamespace(batchsize=64), dp=types.SimpleNamespace(world=2))
    with pytest.raises(ValueError, match='fewer slices than one global batch'):
        AEMODEL._num_batches(tr, Split(), Phase.VAL)
    tr.dp.world = 1
    assert AEMODEL._num_batches(tr, Split(), 'VAL') == 3


def test_every_environment_switch_is_documented():
    """Every UAD_* switch the library, the package or bench.py reads appears in README.md's switch section."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'unsupervised_anomaly_detection_brain_mri_amd')
    names = set()
    for f in glob.glob(os.path.join(pkg, 'csrc', '*')):
        names |= set(re.findall(r'getenv\("(UAD_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(pkg, '*.py')) + glob.glob(os.path.join(pkg, '*', '*.py')) + [os.path.join(root, 'bench.py'), os.path.join(root, 'run.py')]:
        names |= set(re.findall(r"environ(?:\.get)?\(?\[?'(UAD_[A-Z0-9_]+)'", open(f).read()))
    readme = open(os.path.join(root, 'README.md')).read()
    missing = sorted(n for n in names if n not in readme)
    assert not missing, f'undocumented switches: {missing}'


def test_bench_conv_tags_from_the_tensor_table():
    """bench.spec_conv_tags (the spatial GMVAE line's roofline: FLOP / algorithmic bytes of a launch group from the handle's tensor table) against the VAE's
    closed forms (bench.flops_per_tag / bytes_per_tag) and SURVEY.md 8d's per-slice figure of the 256 x 256 restoration pass."""
    import bench
    from oracle import gmvae as og
    from oracle import vae as ov
    n = 64
    tags = bench.spec_conv_tags(ov.param_spec('VAE', 128, 128, 1, 8, 128), 128, n)
    fl, by = bench.flops_per_tag(n), bench.bytes_per_tag(n, fin_bits=False)
    assert set(tags) == set(fl)
    for t in fl:
        assert tags[t][0] == fl[t], t
        if t != 'dec3.fwd':                      # (the fused last block also reads the target and writes x_hat / L1: 12 B per pixel more)
            assert tags[t][1] == by[t], t
    assert by['dec3.fwd'] - tags['dec3.fwd'][1] == 12.0 * n * 128 * 128
    g = bench.spec_conv_tags(og.param_spec(256, 256, 1, 8, 9, 1, 1), 256, 1)
    per_slice = sum(v[0] for k, v in g.items() if k.endswith('.fwd')) + sum(v[0] for k, v in g.items() if k.endswith('.dgrad') and k != 'enc0.dgrad')
    assert abs(per_slice / 4.88e9 - 1.0) < 0.01          # SURVEY.md 8d counts the 1 x 1 heads and the final conv too (0.6 %)


def test_bench_evidence_loader_and_stdout_claim(tmp_path):
    """bench.py (round 5): a line's `roofline.traffic` / `roofline.rocprof` for the workloads other than the default one come from profiles/r06_evidence_<workload>.json (r05_ as fallback)
    (tools/evidence.py) through the kernel template of the LAST decoder block's launch groups; and rank 0's one JSON line is written to the real stdout while
    everything else -- RCCL's start-up banner on fd 1 included -- goes to stderr."""
    import json
    import subprocess
    import sys
    import bench
    assert bench.last_block_kernel('dec3.fwd', 4) == 'conv5_d16s_kernel<16, 16, 32, 4, 1, 2, 2' and bench.last_block_kernel('dec3.fwd', 4, 'bf16x6') == 'conv5_d16s_kernel<8, 16, 32, 4, 1, 1, 2'
    assert bench.last_block_kernel('dec4.dgrad', 5).startswith('conv5_f16_kernel<8, 16, 16')
    assert bench.last_block_kernel('dec3.wgrad', 4, 'f32') == 'conv5_w_kernel'
    assert bench.last_block_kernel('dec2.fwd', 4) is None and bench.last_block_kernel('enc1.fwd', 4) is None          # several layers share those templates
    for wl, tag, nblk in (('cevae_b16', 'dec3.wgrad', 4), ('gmvae_restore_b16', 'dec4.fwd', 5)):
        flop = 13.42e9
        tr, rp = bench.evidence_for(wl, bench.last_block_kernel(tag, nblk), flop, 2500.0 / 3)
        assert tr and rp, wl
        assert tr['bytes'] == tr['fetch_bytes'] + tr['write_bytes'] and 2e7 < tr['bytes'] < 3e8, (wl, tr)
        assert 0.05 < rp['frac'] < 0.6 and rp['calls'] > 0 and '_evidence_' + wl in rp['source'], (wl, rp)
    assert bench.evidence_for('no_such_workload', 'x', 1.0, 1.0) == (None, None) and bench.evidence_for('cevae_b16', None, 1.0, 1.0) == (None, None)
    # fd 1 is protected: a child that claims stdout, then writes to fd 1 natively (as RCCL does) and prints, still leaves exactly the JSON line on stdout
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); os.write(1, b'RCCL version : banner\\n'); print('chatter'); "
            "bench.emit_json({'metric': 'm', 'value': 1})" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {'metric': 'm', 'value': 1} and r.stdout.count('\n') == 1
    assert 'banner' in r.stderr and 'chatter' in r.stderr


def test_data_parallel_step_prefers_the_library_path_only_under_rccl(monkeypatch):
    """parallel._library_allreduce_default: the library-issued all-reduce is the default only when the process group's backend is RCCL ("nccl") and
    UAD_DP_LIBRARY_AR is not 0; gloo (every CPU test) and an uninitialised group keep the torch.distributed path."""
    from unsupervised_anomaly_detection_brain_mri_amd import parallel
    import torch.distributed as dist
    monkeypatch.delenv('UAD_DP_LIBRARY_AR', raising=False)
    monkeypatch.setattr(dist, 'is_initialized', lambda: False)
    assert parallel._library_allreduce_default() is False
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_backend', lambda *a, **k: 'gloo')
    assert parallel._library_allreduce_default() is False
    monkeypatch.setattr(dist, 'get_backend', lambda *a, **k: 'nccl')
    assert parallel._library_allreduce_default() is True
    monkeypatch.setenv('UAD_DP_LIBRARY_AR', '0')
    assert parallel._library_allreduce_default() is False
    # the bucket plan handed to uad_allreduce_attach: contiguous slices, every gradient element in exactly one bucket
    segs = {parallel._lib.SEG_DECODER: (1000, 700), parallel._lib.SEG_BOTTLENECK: (600, 400), parallel._lib.SEG_ENCODER_HI: (50, 550), parallel._lib.SEG_ENCODER_LO: (0, 50)}
    for b in (4, 3, 2, 1):
        plan = parallel.bucket_plan(segs, b)
        assert len(plan) == b and sum(c for _, _, c in plan) == 1700
        covered = sorted((o, o + c) for _, o, c in plan)
        assert covered[0][0] == 0 and covered[-1][1] == 1700 and all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))
