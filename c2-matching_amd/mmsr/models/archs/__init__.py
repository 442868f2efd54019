"""Arch registry with the reference's semantics (archs/__init__.py:9-18): every ``*_arch.py`` next to this file is
imported and searched by class name.  No mmcv needed (``mmcv.scandir`` was only used to list the directory)."""
import importlib
import os

_folder = os.path.dirname(os.path.abspath(__file__))
arch_filenames = sorted(os.path.splitext(f)[0] for f in os.listdir(_folder) if f.endswith('_arch.py'))
_arch_modules = [importlib.import_module(f'mmsr.models.archs.{name}') for name in arch_filenames]
