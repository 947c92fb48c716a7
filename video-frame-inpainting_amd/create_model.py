"""``model_key`` -> model (reference src/models/create_model.py:19-111), for the keys on the bi-TAI hot path.

Kept: the named keys ``TAI_gray`` / ``TAI_color``, the models built from the same blocks (MC-Net baseline, bi-TWI, bi-TWA,
bi-SA, TW_P_F: SURVEY.md 8f rank 1), the fallback to a JSON file path and then to an inline JSON string ``{"class",
"args", "kwargs"}`` (:88-111).  The other keys of the reference (SCT, SloMo, optical flow) are different model families
outside this path; asking for one raises with the list of supported keys instead of silently building something else.
"""
import json
import os

from .ablations import (BidirectionalSimpleAverageFillInModel, BidirectionalTimeWeightedAverageFillInModel,
                        TimeWeightedInterpolationFillInModel, TimeWeightedPFFillInModel)
from .mcnet import MCNetFillInModel
from .tai import TAIFillInModel

_BUILDERS = {
    'TAI_gray': lambda: TAIFillInModel(64, 1, 3, 51, num_block=5),          # create_model.py:27-28
    'TAI_color': lambda: TAIFillInModel(64, 3, 3, 51, num_block=4),         # create_model.py:29-30
    'MCNet_gray': lambda: MCNetFillInModel(64, 1, 3),                       # create_model.py:33-34
    'MCNet_color': lambda: MCNetFillInModel(64, 3, 3),                      # create_model.py:35-36
    # ablations on the same blocks (create_model.py:73-86)
    'TimeWeightedInterpolationFillInModel_gray': lambda: TimeWeightedInterpolationFillInModel(64, 1, 3, 51, num_block=5),
    'TimeWeightedInterpolationFillInModel_color': lambda: TimeWeightedInterpolationFillInModel(64, 3, 3, 51, num_block=4),
    'BidirectionalSimpleAverageFillInModel_gray': lambda: BidirectionalSimpleAverageFillInModel(64, 1, 3),
    'BidirectionalSimpleAverageFillInModel_color': lambda: BidirectionalSimpleAverageFillInModel(64, 3, 3),
    'BidirectionalTimeWeightedAverageFillInModel_gray': lambda: BidirectionalTimeWeightedAverageFillInModel(64, 1, 3),
    'BidirectionalTimeWeightedAverageFillInModel_color': lambda: BidirectionalTimeWeightedAverageFillInModel(64, 3, 3),
    'TimeWeightedPFFillInModel': lambda: TimeWeightedPFFillInModel(),
}
_CLASSES = {c.__name__: c for c in (TAIFillInModel, MCNetFillInModel, TimeWeightedInterpolationFillInModel,
                                    BidirectionalSimpleAverageFillInModel, BidirectionalTimeWeightedAverageFillInModel,
                                    TimeWeightedPFFillInModel)}


def supported_model_keys():
    return sorted(_BUILDERS)


def create_model(model_key):
    if model_key in _BUILDERS:
        return _BUILDERS[model_key]()
    print('Could not determine the model to create from key %s. Treating key as file path...' % model_key)
    if os.path.isfile(model_key):
        with open(model_key, 'r') as f:
            return _construct_model_from_dict(json.load(f))
    print('Could not find file %s. Treating key as JSON string...' % model_key)
    try:
        model_info = json.loads(model_key)
    except ValueError:
        raise RuntimeError('Failed to parse model key as a JSON object (named keys on this path: %s)'
                           % ', '.join(supported_model_keys()))
    return _construct_model_from_dict(model_info)


def _construct_model_from_dict(model_info):
    assert isinstance(model_info.get('class'), str)
    assert isinstance(model_info.get('args'), list)
    assert isinstance(model_info.get('kwargs'), dict)
    if model_info['class'] not in _CLASSES:
        raise RuntimeError('Model class %s is not on the bi-TAI hot path (available: %s)'
                           % (model_info['class'], ', '.join(sorted(_CLASSES))))
    return _CLASSES[model_info['class']](*model_info['args'], **model_info['kwargs'])
