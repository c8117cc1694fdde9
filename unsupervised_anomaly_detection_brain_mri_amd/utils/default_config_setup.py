"""utils/default_config_setup.py — options dict and trainer Config fill-in with the reference's keys
(get_options :21-57, get_config :245-271, Dataset enum :13-18).  The MR dataset loaders are out of scope
(no data here): get_datasets returns the synthetic stand-in unless a loader is injected."""
import json
import os
from enum import Enum

from .synthetic import SyntheticDataset, SyntheticPatientDataset

base_path = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Dataset(Enum):
    BRAINWEB = 'BRAINWEBDIR'
    MSSEG2008_UNC = 'MSSEG2008DIR'
    MSISBI2015 = 'MSISBI2015DIR'
    MSLUB = 'MSLUBDIR'


def get_options(batchsize, learningrate, numEpochs, zDim, outputWidth, outputHeight, slices_start=20, slices_end=130,
                numMonteCarloSamples=0, config=None):
    options = {}
    if config:
        options["globals"] = config
    else:
        path = os.path.join(base_path, "config.default.json")
        options["globals"] = json.load(open(path)) if os.path.isfile(path) else {
            "BRAINWEBDIR": "", "MSSEG2008DIR": "", "MSISBI2015DIR": "", "MSLUBDIR": "",
            "CHECKPOINTDIR": "checkpoints", "SAMPLEDIR": "samples"}
    options['debug'] = False
    options['data'] = {}
    options['train'] = {'checkpointDir': options["globals"]["CHECKPOINTDIR"], 'samplesDir': options["globals"]["SAMPLEDIR"],
                        'batchsize': batchsize, 'learningrate': learningrate, 'numEpochs': numEpochs, 'zDim': zDim,
                        'snapshotAfter': 1000, 'outputWidth': outputWidth, 'outputHeight': outputHeight,
                        'useTensorboard': True, 'useMatplotlib': False, 'tensorboardPort': 9001}
    options['sliceStart'] = slices_start
    options['sliceEnd'] = slices_end
    options['threshold'] = 'bestdice'
    options['exportVolumes'] = False
    options['exportPRC'] = True
    options['exportROC'] = True
    options['numMonteCarloSamples'] = numMonteCarloSamples
    options['keepOnlyPositiveResiduals'] = True
    options['applyHyperIntensityPrior'] = True
    options['medianFiltering'] = True
    options['erodeBrainmask'] = True
    return options


def get_datasets(options, dataset=Dataset.BRAINWEB, loader=None):
    if not isinstance(dataset, Dataset):
        raise ValueError(f'No valid dataset given: {dataset}')
    if loader is not None:
        return loader(options, dataset)
    h, w = options['train']['outputHeight'], options['train']['outputWidth']
    # (dataset_hc, dataset_pc) like the reference (:59-242): healthy training set, patient-structured lesion set for the evaluation.  The
    # stand-in lesion set differs per Dataset member (seed, native resolution) so that run.py's per-dataset evaluations are not copies.
    k = list(Dataset).index(dataset)
    sl = max(4, min(16, options['sliceEnd'] - options['sliceStart']))
    pc = SyntheticPatientDataset(n_val=2, n_test=2, slices=sl, native=h + 16 * (k % 2), h=h, w=w, seed=5 + 11 * k, slice_start=0, slice_end=sl)
    return SyntheticDataset(256, 64, h, w, seed=0), pc


def get_config(trainer, options, optimizer, intermediateResolutions, dropout_rate, dataset):
    config = trainer.Config()
    config.dataset = type(dataset).__name__
    config.description = ''
    config.numChannels = dataset.num_channels
    config.batchsize = options['train']['batchsize']
    config.checkpointDir = options['train']['checkpointDir']
    config.snapShotAfter = options['train']['snapshotAfter']
    config.sampleDir = options['train']['samplesDir']
    config.learningrate = options['train']['learningrate']
    config.numEpochs = options['train']['numEpochs']
    config.zDim = options['train']['zDim']
    config.beta1 = 0.5
    config.outputHeight = options['train']['outputHeight']
    config.outputWidth = options['train']['outputWidth']
    config.useTensorboard = options['train']['useTensorboard']
    config.useMatplotlib = options['train']['useMatplotlib']
    config.tensorboardPort = options['train']['tensorboardPort']
    config.debugGradients = options['debug']
    config.optimizer = optimizer
    config.intermediateResolutions = intermediateResolutions
    config.weightRegularization = 0.0
    config.dropout_rate = dropout_rate
    config.dropout = False
    config.l1_weight = 1.0
    config.options = options
    return config
