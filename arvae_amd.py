"""Import alias for the product package.

The package directory is called ``ar-vae_amd`` (the project's name), which is
not a valid Python identifier.  ``import arvae_amd`` loads that directory as a
regular package under the importable name ``arvae_amd``; sub-modules resolve
through its ``__path__`` (``import arvae_amd.ops`` -> ``ar-vae_amd/ops.py``).
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ar-vae_amd")
_spec = importlib.util.spec_from_file_location(
    __name__,
    os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
