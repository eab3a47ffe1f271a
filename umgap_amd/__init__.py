"""umgap_amd — MI355X-native implementation of UMGAP's pept2lca -> taxa2agg hot path.

The compute lives in libumgap_hip.so (hand-written HIP for gfx950, C ABI in include/umgap_hip.h);
this package is the thin Python mirror of the reference interface used by tests and bench.py.
Importing the package does not load the library; the first use does, and fails loudly if the
shared object has not been built (there is no CPU fallback).
"""
from ._lib import UmgapError, declared_symbols, load  # noqa: F401

__all__ = ["UmgapError", "declared_symbols", "load"]
