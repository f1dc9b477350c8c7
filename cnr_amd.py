"""Importable alias of the package directory ``category-nerf-reconstruction-official_amd/`` (its name
carries a hyphen, which Python cannot import directly):  ``import cnr_amd`` loads that directory as
the package ``cnr_amd`` so ``from cnr_amd import trainer, loss`` etc. work."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "category-nerf-reconstruction-official_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
