"""Model registry with the reference's semantics (models/__init__.py:11-43): every ``*_model.py`` here is imported and
``create_model(opt)`` instantiates ``opt['model_type']`` by class name."""
import importlib
import logging
import os

_folder = os.path.dirname(os.path.abspath(__file__))
model_filenames = sorted(os.path.splitext(f)[0] for f in os.listdir(_folder) if f.endswith('_model.py'))
_model_modules = [importlib.import_module(f'mmsr.models.{name}') for name in model_filenames]


def create_model(opt):
    model_type = opt['model_type']
    for module in _model_modules:
        model_cls = getattr(module, model_type, None)
        if model_cls is not None:
            model = model_cls(opt)
            logging.getLogger('base').info(f'Model [{model.__class__.__name__}] is created.')
            return model
    raise ValueError(f'Model {model_type} is not found.')
