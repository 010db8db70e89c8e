"""MI355X-native neural Viterbi decode path (feed-forward decoder).

Host-side mirror of the reference's Python surface for this one path
(``fnn_viterbi`` / ``fnn_viterbi_mixed_snr``) over a hand-written HIP kernel
for gfx950 reached through the C ABI in ``include/nnviterbi.h``.

The directory name is not a Python identifier; import it through the
``nnviterbi_amd`` loader at the repository root.
"""
from ._cabi import NNViterbiError, build_library, library_path, load_library  # noqa: F401

__all__ = ["NNViterbiError", "build_library", "library_path", "load_library"]
