"""Import alias for the package directory ``neural-network-based-viterbi-decoding_amd``.

That directory name (fixed by the project layout) contains hyphens and so
cannot be imported by name; ``import nnviterbi_amd`` loads it as a regular
package under this name, submodules included
(``nnviterbi_amd.fnn_viterbi``, ``nnviterbi_amd.decoder`` ...).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "neural-network-based-viterbi-decoding_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
