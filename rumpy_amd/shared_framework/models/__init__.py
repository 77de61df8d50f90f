"""Model registry - the plugin boundary of rumpy/shared_framework/models/__init__.py:7-35.

Every class ``<Name>Handler`` found (by AST scan, at import time) in ``<pkg>/<task>/models/<category>/handlers.py``
is registered under ``name.lower()``; ``define_model(name, **kwargs)`` instantiates it.
"""
import ast
import os
from pydoc import locate

code_base_directory = os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir, os.pardir))
package_name = os.path.basename(code_base_directory)

ml_tasks = ['SISR', 'VSR', 'regression']
available_models = {}

for _task in ml_tasks:
    _model_dir = os.path.join(code_base_directory, _task, 'models')
    if not os.path.isdir(_model_dir):
        continue
    for _entry in sorted(os.scandir(_model_dir), key=lambda e: e.name):
        if not _entry.is_dir() or '__' in _entry.name:
            continue
        _handler_file = os.path.join(_model_dir, _entry.name, 'handlers.py')
        if not os.path.isfile(_handler_file):
            continue
        with open(_handler_file, 'r') as _f:
            _tree = ast.parse(_f.read())
        for _node in ast.walk(_tree):
            if isinstance(_node, ast.ClassDef):
                available_models[_node.name.split('Handler')[0].lower()] = \
                    '%s.%s.models.%s.handlers.%s' % (package_name, _task, _entry.name, _node.name)


def define_model(name, **kwargs):
    """Instantiate the handler registered under ``name`` (KeyError for unknown names, like the reference)."""
    return locate(available_models[name])(**kwargs)
