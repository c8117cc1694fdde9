"""GPU: run.py end to end (reference CLI, run.py:18-152) -- train on the synthetic healthy set, then the reference's evaluation flow
(evaluate_optimal / determine_threshold_on_labeled_patients / evaluate_with_threshold, run.py:58-116) through Evaluation.evaluate on the
patient-structured stand-in datasets; evalPC files land under <SAMPLEDIR>/<network>/<model_dir>/eval-<epoch>-<timestamp>-<description>/."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(tmp_path, *extra):
    import run
    cfg = tmp_path / 'config.json'
    cfg.write_text(json.dumps({"BRAINWEBDIR": "", "MSSEG2008DIR": "", "MSISBI2015DIR": "", "MSLUBDIR": "",
                               "CHECKPOINTDIR": str(tmp_path / 'ck'), "SAMPLEDIR": str(tmp_path / 'smp')}))
    return run.build_parser().parse_args(['-c', str(cfg), '-b', '8', '-E', '1', '-z', '64', '-w', '64', '-g', '64', '-e', '32', *extra])


def _evals(tmp_path):
    return sorted(glob.glob(str(tmp_path / 'smp' / '*' / '*' / 'eval-*')))


def test_full_flow_without_threshold(tmp_path, capsys):
    import run
    run.main(_args(tmp_path, '-t', 'VAE', '-m', 'variational_autoencoder'))
    dirs = _evals(tmp_path)
    names = [os.path.basename(d).split('-', 4)[-1] for d in dirs]
    # 3 datasets x {without, with prior} best-dice evaluations, the VAL threshold directory, 3 fixed-threshold evaluations (run.py:58-97)
    assert sum('_upperbound_bestdice' in n and not n.endswith('_wPrior') for n in names) == 3
    assert sum(n.endswith('_upperbound_bestdice_wPrior') for n in names) == 3
    assert sum('VALthresh_' in n for n in names) == 3 and any(n.endswith('VAL') for n in names)
    ev = np.load(os.path.join([d for d in dirs if 'MSISBI2015-VALthresh_' in d][0], 'evalPC.npy'), allow_pickle=True).item()
    assert ev['thresholdType'] != 'bestdice' and 0.0 <= ev['diff_AUPRC'] <= 1.0 and len(ev['DiceScorePerPatient']) == 2
    out = capsys.readouterr().out
    assert 'Optimal threshold on MS Lesion Validation Set' in out and json.loads(out.strip().splitlines()[-1])['diff_AUC'] >= 0.0


@pytest.mark.parametrize('trainer,model', [('ceVAE', 'context_encoder_variational_autoencoder'), ('AE', 'autoencoder')])
def test_single_dataset_and_fixed_threshold(tmp_path, trainer, model):
    import run
    run.main(_args(tmp_path, '-t', trainer, '-m', model, '-d', 'MSLUB'))                      # one best-dice evaluation, then return (run.py:63-67)
    d1 = _evals(tmp_path)
    assert len(d1) == 1 and 'MSLUB_upperbound_bestdice_wPrior' in d1[0] and os.path.isfile(os.path.join(d1[0], 'evalPC.txt'))
    run.main(_args(tmp_path, '-t', trainer, '-m', model, '-d', 'BRAINWEB', '-O', '0.05'))     # resumes the checkpoint; threshold + dataset (run.py:85-86)
    d2 = [d for d in _evals(tmp_path) if d not in d1]
    assert len(d2) == 1 and 'BRAINWEB-VALthresh_0.05' in d2[0]
    ev = np.load(os.path.join(d2[0], 'evalPC.npy'), allow_pickle=True).item()
    assert ev['thresholdType'] == 0.05 and np.isfinite(ev['DiceScore'])
    with pytest.raises(SystemExit):
        run.main(_args(tmp_path, '-t', trainer, '-m', model, '-d', 'Brainweb2'))
