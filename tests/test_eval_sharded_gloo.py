"""CPU, world_size 2 over gloo: patient-sharded evaluation (SURVEY.md 8e "Inference / config 5"; utils/Evaluation.py:183-365,416-461 is the
reference's single-process loop).  utils.Evaluation shards the per-patient walk over the ranks of an initialised process group, exchanges the
finished residual volumes and scores the SAME ordered patient list on every rank -- so `evaluate`, `determine_threshold_on_labeled_patients`
and the array-level `evaluate_arrays` must return, on both ranks, exactly what a single process returns: bit-exact integer counts, scores
within 1e-6 (here: equal), and each rank must have reconstructed only its own patients.  The model is the CPU stand-in of
tests/test_evaluation_entry.py (host scoring ops = the reference-pinned scoring oracle), so no GPU is needed."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

KEYS = ('diff_AUC', 'diff_AUPRC', 'bestDiceScore', 'bestThreshold', 'DiceScore', 'DiceScorePerPatient', 'PrecisionPerPatient', 'RecallPerPatient',
        'l1reconstructionErrorMean', 'l1reconstructionErrorVariance')


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run_all(tmp, world):
    """What one process (or one rank) computes: the three entry points on a 5-patient TEST / 3-patient VAL synthetic set."""
    import pathlib
    from tests.test_evaluation_entry import BlurModel, _opts
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticPatientDataset
    tmp = pathlib.Path(tmp)
    opt = _opts(tmp)
    ds = SyntheticPatientDataset(n_val=3, n_test=5, slices=12, native=80, h=64, w=64, seed=1, slice_start=0, slice_end=12)
    model = BlurModel(tmp)
    ev = Evaluation.evaluate(ds, model, opt, epoch='1', description=f'w{world}')
    res = {k: ev[k] for k in KEYS}
    res['files'] = sorted(os.listdir(ev['eval_dir']))
    res['patients_reconstructed'] = len(model.calls) // 3              # 12 slices in batches of 5 = 3 calls per patient
    model.calls.clear()
    res['val'] = Evaluation.determine_threshold_on_labeled_patients([ds], model, opt, description='VAL')
    res['val_patients_reconstructed'] = len(model.calls) // 3
    # array-level entry point with Monte-Carlo dropout records riding through the exchange
    vols, labs, masks = [], [], []
    for k in ds.get_patient_idx('TEST'):
        p = ds.patients[k]
        x, seg, skull, _, _ = Evaluation.collect_patient_volume(ds, p, p['filtered_files'][0], opt)
        vols.append(x); labs.append(seg); masks.append(skull)
    ea = Evaluation.evaluate_arrays(vols, labs, masks, model, opt, eps=0.0)
    res['arrays'] = {k: ea[k] for k in KEYS}
    # integer evidence: per-patient voxel counts of the thresholded, component-filtered prediction
    res['n_label_voxels'] = [int(np.count_nonzero(l)) for l in labs]
    return res


def _worker(rank, world, port, tmp, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _run_all(os.path.join(tmp, f'rank{rank}'), world)))
    finally:
        dist.destroy_process_group()


def test_patient_sharded_evaluation_equals_single_process(tmp_path):
    ref = _run_all(str(tmp_path / 'single'), 1)
    assert ref['patients_reconstructed'] == 5 and ref['val_patients_reconstructed'] == 3
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        r = got[rank]
        for k in KEYS:                       # every rank holds the single-process result
            assert np.array_equal(np.asarray(r[k], np.float64), np.asarray(ref[k], np.float64), equal_nan=True), (rank, k, r[k], ref[k])
            assert np.array_equal(np.asarray(r['arrays'][k], np.float64), np.asarray(ref['arrays'][k], np.float64), equal_nan=True), (rank, 'arrays', k)
        assert r['val'] == ref['val'] and r['n_label_voxels'] == ref['n_label_voxels']
        # each rank reconstructed only its own patients: k mod world == rank of 5 TEST / 3 VAL patients
        assert r['patients_reconstructed'] == len(range(rank, 5, world)) and r['val_patients_reconstructed'] == len(range(rank, 3, world))
    # rank 0 alone writes the evaluation files
    assert {'evalPC.npy', 'evalPC.txt'} <= set(got[0]['files']) and not ({'evalPC.npy', 'evalPC.txt'} & set(got[1]['files']))
