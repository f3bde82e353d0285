"""miekki_amd -- MI355X (gfx950) implementation of Miekki's sketch-build and
fingerprint-intersection hot path.

The product is `libmiekki_hip.so` (hand-written HIP kernels behind the C ABI of
include/miekki_hip.h) and the `miekki` host binary.  This package is the thin
Python mirror of the reference's Miekki class used by the tests and bench.py; it
only ever calls the C ABI -- there is no CPU path.
"""
from .lib import MiekkiHipError, load_library, library_path  # noqa: F401
from .index import Miekki, SimilarityScore  # noqa: F401
