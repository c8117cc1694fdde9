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
    if args.threshold:
        options['threshold'] = args.threshold
    if args.cache:
        # a real-data slice cache has no per-dataset loaders behind it: score its TEST patients once (array-level core)
        from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import volumes_from_cache
        vols, labs, masks, _ = volumes_from_cache(args.cache, 'TEST')
        ev = Evaluation.evaluate_arrays(vols, labs, masks, model, options)
        print(json.dumps(_summary(ev), default=float))
        return

    def ds_of(name):
        if isinstance(name, Dataset):
            return name
        try:
            return Dataset[str(name).upper()]            # the reference passes the raw -d string on, which cannot work (run.py:67); accept the member's name
        except KeyError:
            raise SystemExit(f'-d {name!r}: expected one of {[d.name for d in Dataset]}')

    results = []
    ########################
    #  Evaluate best dice  #  (run.py:58-80)
    ########################
    if not args.threshold:
        if args.ds:
            results.append(evaluate_optimal(model, options, ds_of(args.ds)))
            print(json.dumps(_summary(results[-1]), default=float))
            return
        for prior in (False, True):     # all datasets for best dice without, then with, the hyper-intensity prior
            options['applyHyperIntensityPrior'] = prior
            for d in (Dataset.BRAINWEB, Dataset.MSLUB, Dataset.MSISBI2015):
                results.append(evaluate_optimal(model, options, d))
    ###############################################
    #  Evaluate generalization to other datasets  #  (run.py:82-97)
    ###############################################
    if args.threshold and args.ds:  # only threshold is invalid
        results.append(evaluate_with_threshold(model, options, args.threshold, ds_of(args.ds)))
    else:
        options['applyHyperIntensityPrior'] = False
        dataset_brainweb = get_evaluation_dataset(options, Dataset.BRAINWEB)
        best_dice_val, thresh_val = Evaluation.determine_threshold_on_labeled_patients([dataset_brainweb], model, options, description='VAL')
        print(f"Optimal threshold on MS Lesion Validation Set without optimal postprocessing: {thresh_val} (Dice-Score {best_dice_val})")
        for d in (Dataset.BRAINWEB, Dataset.MSLUB, Dataset.MSISBI2015):      # re-evaluate with the previously determined threshold
            results.append(evaluate_with_threshold(model, options, thresh_val, d))
    print(json.dumps(_summary(results[-1]), default=float))


def _summary(ev):
    return {k: (v if isinstance(v, (list, dict, str)) else float(v)) for k, v in ev.items() if k not in ('time', 'epistemic_variance')}


def get_evaluation_dataset(options, dataset):          # run.py:115-117
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_datasets
    options['data']['dir'] = options["globals"].get(dataset.value, '')
    return get_datasets(options, dataset=dataset)[1]


def evaluate_with_threshold(model, options, threshold, dataset):          # run.py:100-105
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    options['applyHyperIntensityPrior'] = False
    options['threshold'] = threshold
    evaluation_dataset = get_evaluation_dataset(options, dataset)
    description = f'{type(evaluation_dataset).__name__}-{dataset.name}-VALthresh_{options["threshold"]}'
    return Evaluation.evaluate(evaluation_dataset, model, options, description=description, epoch=str(options['train']['numEpochs']))


def evaluate_optimal(model, options, dataset):          # run.py:108-116
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    prior_str = "_wPrior" if options['applyHyperIntensityPrior'] else ''
    evaluation_dataset = get_evaluation_dataset(options, dataset)
    description = f'{type(evaluation_dataset).__name__}-{dataset.name}_upperbound_{options["threshold"]}{prior_str}'
    return Evaluation.evaluate(evaluation_dataset, model, options, description=description, epoch=str(options['train']['numEpochs']))


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
