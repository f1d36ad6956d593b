"""Model registry with the reference's `load_model(name)` (FCGF_APR/model/__init__.py:18-32)."""
import logging

from . import resunet as resunets

MODELS = []


def add_models(module):
    MODELS.extend([getattr(module, a) for a in dir(module) if 'Net' in a or 'MLP' in a])


add_models(resunets)


def load_model(name):
    """Return the model class with that name (None + a log line if unknown, as the reference does)."""
    mdict = {model.__name__: model for model in MODELS}
    if name not in mdict:
        logging.info(f'Invalid model index. You put {name}. Options are:')
        for model in MODELS:
            logging.info('\t* {}'.format(model.__name__))
        return None
    return mdict[name]
