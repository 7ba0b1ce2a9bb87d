"""Import shim.  The package directory is `madnlp.jl_amd/` (named after the reference
repository, MadNLP.jl); a dotted directory name is not importable, so this module loads
that directory as the package `madnlp_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "madnlp.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
