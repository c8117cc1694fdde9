#!/usr/bin/env python
"""run.py — the reference's CLI driver (run.py:18-152) on the MI355X-native path: same flags, same trainer/model
lookup BY NAME, same train -> evaluate flow.  Fixed consciously: Dataset.Brainweb -> Dataset.BRAINWEB (A12) and the
tuple / bool flag parsing (-i, -G).  Without real MR data (none can be downloaded here) it runs on the synthetic
Brainweb-like dataset."""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PKG = 'unsupervised_anomaly_detection_brain_mri_amd'


def main(args):
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import (
        get_config, get_options, get_datasets, Dataset)
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
    try:
        trainer = getattr(importlib.import_module(f'{PKG}.trainers.{args.trainer}'), args.trainer)
        network = getattr(importlib.import_module(f'{PKG}.models.{args.model}'), args.model)
    except (ImportError, AttributeError) as e:
        raise SystemExit(f'trainer {args.trainer!r} / model {args.model!r} is not on the MI355X path: {e}')
    json_config = None
    if args.config and os.path.isfile(args.config):
        with open(args.config) as f:
            json_config = json.load(f)
    options = get_options(batchsize=args.batchsize, learningrate=args.lr, numEpochs=args.numEpochs, zDim=args.zDim,
                          outputWidth=args.outputWidth, outputHeight=args.outputHeight, slices_start=args.slices_start,
                          slices_end=args.slices_end, numMonteCarloSamples=args.numMonteCarloSamples, config=json_config)
    if args.cache:
        # a slice cache built from real volumes (tools/build_cache.py, utils/nifti.py): HBM-resident training set, per-patient TEST volumes
        from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import DeviceDataset
        dataset_hc = DeviceDataset.from_cache(args.cache)
        if tuple(dataset_hc.shape[1:3]) != (args.outputHeight, args.outputWidth):
            raise SystemExit(f'cache slices are {dataset_hc.shape[1]}x{dataset_hc.shape[2]}, but -g/-w ask for {args.outputHeight}x{args.outputWidth}')
    else:
        dataset_hc, dataset_pc = get_datasets(options, dataset=Dataset.BRAINWEB)
    config = get_config(trainer=trainer, options=options, optimizer=args.optimizer,
                        intermediateResolutions=args.intermediateResolutions, dropout_rate=0.2, dataset=dataset_hc)
    for arg in vars(args):
        if hasattr(config, arg):
            setattr(config, arg, getattr(args, arg))
    model = trainer(None, config, network=network)
    model.train(dataset_hc)
    # evaluation on synthetic lesion volumes (Evaluation.evaluate; Brainweb/MSLUB/MSISBI2015 need the real data)
    vols, labs, masks = [], [], []
    if args.cache:
        from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import volumes_from_cache
        vols, labs, masks, _ = volumes_from_cache(args.cache, 'TEST')
    for p in range(0 if args.cache else 2):
        x, lab, msk = synthetic_slices(16, args.outputHeight, args.outputWidth, seed=50 + p, lesions=True)
        vols.append(x[..., 0].astype('float64')); labs.append(lab); masks.append(msk)
    if args.threshold:
        options['threshold'] = args.threshold
    ev = Evaluation.evaluate(vols, labs, masks, model, options)
    summary = {k: (v if isinstance(v, (list, dict, str)) else float(v)) for k, v in ev.items() if k not in ('time', 'epistemic_variance')}
    # evalPC.npy / evalPC.txt under <SAMPLEDIR>/<network>/<model_dir>/eval-<epoch>-<timestamp>/ (utils/Evaluation.py:380-395,519-526)
    try:
        import time
        import numpy as np
        epoch = len(model.curves.get('TRAIN/loss', model.curves.get('TRAIN/reconstructionLoss', [])))
        out_dir = os.path.join(options['train']['samplesDir'], network.__name__, model.model_dir, f"eval-{epoch}-{time.strftime('%Y%m%d_%H%M%S')}")
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, 'evalPC.npy'), summary)
        with open(os.path.join(out_dir, 'evalPC.txt'), 'w') as f:
            json.dump(summary, f, default=float)
    except Exception as e:      # the summary on stdout is the contract; the files are a convenience
        print(f'could not write the evaluation summary: {e}')
    print(json.dumps(summary, default=float))


def build_parser():
    ap = argparse.ArgumentParser(description='Framework')
    ap.add_argument('-c', '--config', default='config.default.json', type=str, help='config-path')
    ap.add_argument('-b', '--batchsize', default=8, type=int)
    ap.add_argument('-l', '--lr', default=0.0001, type=float)
    ap.add_argument('-E', '--numEpochs', default=1000, type=int)
    ap.add_argument('-z', '--zDim', default=128, type=int)
    ap.add_argument('-w', '--outputWidth', default=128, type=int)
    ap.add_argument('-g', '--outputHeight', default=128, type=int)
    ap.add_argument('-o', '--optimizer', default='ADAM', type=str, help='ADAM | SGD | MOMENTUM | RMS (the non-Adam rules run on the fused AE-family handle; the WGAN trainers build their own Adams)')
    ap.add_argument('-i', '--intermediateResolutions', default=(8, 8), type=lambda s: tuple(int(v) for v in s.split(',')))
    ap.add_argument('-s', '--slices_start', default=20, type=int)
    ap.add_argument('-e', '--slices_end', default=130, type=int)
    ap.add_argument('-t', '--trainer', default='AE', type=str)
    ap.add_argument('-m', '--model', default='autoencoder', type=str)
    ap.add_argument('-O', '--threshold', default=None, type=float)
    ap.add_argument('-d', '--ds', default=None, type=str)
    ap.add_argument('-n', '--numMonteCarloSamples', default=0, type=int)
    ap.add_argument('--cache', default=None, type=str, help='slice-cache directory (tools/build_cache.py) to train / evaluate on instead of the synthetic set')
    # GMVAE-only flags (reference run.py:144-150, same defaults)
    ap.add_argument('-C', '--dim_c', default=9, type=int, help='only for GMVAE')
    ap.add_argument('-Z', '--dim_z', default=128, type=int, help='only for GMVAE')
    ap.add_argument('-W', '--dim_w', default=1, type=int, help='only for GMVAE')
    ap.add_argument('-A', '--c_lambda', default=1, type=int, help='only for GMVAE')
    ap.add_argument('-L', '--restore_lr', default=1e-3, type=float, help='only for GMVAE')
    ap.add_argument('-S', '--restore_steps', default=150, type=int, help='only for GMVAE')
    ap.add_argument('-T', '--tv_lambda', default=-1.0, type=float, help='only for GMVAE')
    return ap


if __name__ == '__main__':
    main(build_parser().parse_args())
